// sbv_api.hip — C-ABI of libsbv.so (include/sbv.h): context, HBM buffers, launches.
//
// One context per process.  HBM layout (device `dev`, sized for `cap` tuples, grown on demand):
//   tuples    cap * 160 B            staging for the host-pointer entry point (AoS, as received)
//   scratch   6 x cap * 32 B + cap   limb-major r, u1, u2, Qx, Qy, sM + ok flags   (stage A -> B)
//   qtab      cap * 1280 B           per-signature window tables k*Q, k = 1..8      (stage B)
//   gtab      17 * 32768 * 64 B      fixed-base 16-bit comb k * 2^(16j) * G (35.7 MB, read-only)
//   bitmap    cap / 8 B
// For the 2^20-tuple headline batch that is 0.17 + 0.20 + 1.34 GB: sized for 288 GB of HBM3E,
// not for a cache.  There is no CPU verification path in this library.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <functional>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <random>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/sbv.h"
#include "ed25519_kernels.h"
#include "p256_kernels.h"
#include "k256_core.h"

namespace {

using sbv::u32;

// staging of the pipelined host-pointer entry (sbv_p256_verify_batch): two of them, so that one call's upload overlaps
// another's kernels
struct StageSlot {
    uint8_t* d_tuples = nullptr; uint8_t* d_bitmap = nullptr; uint8_t* h_bitmap = nullptr;
    size_t cap = 0;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // H2D start / end, stage A end, stage B end, D2H end
    bool used = false;
};

struct Context {
    std::mutex mu;                      // held for the whole of every call that touches this device's buffers
    bool ready = false;
    int device = -1;                    // index of this context: what sbv_init / the sharded entries call a device (g_ctxs, g_shard, g_part)
    int hip_dev = -1;                   // the HIP device behind it: == device unless SBV_LOGICAL_DEVICES folds several contexts onto one GPU
    int mem_share = 1;                  // contexts planned on that HIP device: every pool budget is divided by it
    size_t cap = 0;
    uint8_t* d_tuples = nullptr;
    uint8_t* d_scratch = nullptr;
    u32* d_qtab = nullptr;
    sbv::apt* d_gtab = nullptr;
    sbv::apt* d_g16r = nullptr;          // the comb of G for the carry-free kernels (R = 2^261 domain, g_bits-wide windows)
    int g_bits = 16;
    uint8_t* d_bitmap = nullptr;
    uint8_t* d_rerun = nullptr;         // per-wavefront flags between the fast and the exact stage-B pass
    uint8_t* h_bitmap = nullptr;        // pinned
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // H2D of the pipelined host-pointer entry
    StageSlot stage[2];
    std::condition_variable slot_cv;
    sbv::GroupSync gsync;               // side streams + events of the grouped stage B
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t busy = nullptr;          // completion of the last launch that used the scratch
    bool busy_valid = false;
    sbv_timing timing{};
    // in-step key grouping (p256_group.h)
    sbv::GroupBuffers grp;
    sbv::EdGroupBuffers edgrp;          // Ed25519 grouped step: per-batch combs of -A (the rest is shared with grp)
    bool group_enabled = true;
    // Batches from this size on take the grouped step.  With the key-table cache ON (the default) that is nearly every batch:
    // cached keys are grouped whatever their count (p256_group.h: group_assign_lane), so a warm batch of a few thousand
    // tuples runs the comb phases (2^12: 0.34 ms against 1.25 ms through the doubling kernel, profiles/r03/
    // sweep_sizes_r03a.jsonl) and a cold one leaves its tables behind for the next.  With the cache OFF nothing outlives
    // the call, and below ~2^17 tuples building tables costs more latency (~2 ms) than the doubling kernel takes (1.3-1.5 ms):
    // group_min_batch_cold applies then.  sbv_p256_set_grouping(min_batch != 0) sets both.
    size_t group_min_batch = 64, group_min_batch_cold = (size_t)1 << 17;
    size_t group_min_batch_ed = (size_t)1 << 18;     // Ed25519 with its key-table cache off: the cold crossover (round 2); with it on, group_min_batch
    size_t group_min_batch_k256 = (size_t)1 << 17;   // secp256k1 with its key-table cache off; with it on, group_min_batch
    // Round 5: a key earns a table (its ROWS: a comb with 4-bit windows) from ~16 uses in a batch — soft threshold, sampled — and the
    // fill that makes it a full 8-bit comb from group_full_min uses; up to 65 536 groups per batch (p256_group.h: table classes).
    // group_min_count = 0: the built-in default — 8 uses counted on every 4th tuple for the P-256 step (tables in classes), 64 for the
    // Ed25519 / secp256k1 steps (always full tables)
    u32 group_min_count = 0, group_max = 65536, group_full_min = 256;
    int group_sample_shift = SBV_GROUP_SAMPLE_SHIFT_DEFAULT;     // SBV_GROUP_SAMPLE_SHIFT (0..6): the P-256 default threshold's sampling rate
    // hot keys (p256_group.h): wide combs for cache slots that keep being hit — how many (0 = off; 35.7 MB each) and from how many tuples on
    u32 hot_keys = 1024, hot_min_hits = 4096;
    u32 ed_hot_keys = 1024, ed_hot_min_hits = 4096;
    bool pools_shrunk = false;          // fit_group_pools() gave this device smaller pools than asked for (sbv_p256_pool_stats)
    unsigned group_nomem_events = 0;    // how often a grouped batch fell back to the one-lane kernel for lack of memory (sbv_p256_pool_stats)
    unsigned group_nomem_skip = 0;      // grouped batches to run ungrouped before the pools are tried again (after an SBV_ENOMEM)
    // persistent key-table caches, one per scheme (SBV_SCHEME_*: P-256, secp256k1, Ed25519; p256_group.h): on / off and
    // cached keys (270 KiB of HBM per ECDSA key, 384 KiB per Ed25519 key)
    bool kc_on[3] = {true, true, true};
    u32 kc_caps[3] = {16384, 1024, 1024};
    sbv::KeyPool k256pool;              // secp256k1: comb pool + key-table cache of its own
    // message front end staging (grown on demand)
    uint8_t* d_msgs = nullptr; size_t msgs_cap = 0;
    uint8_t* d_sigs = nullptr; size_t sigs_cap = 0;
    uint64_t* d_moff = nullptr; uint64_t* d_soff = nullptr; size_t moff_cap = 0, soff_cap = 0;
    sbv::aniels* d_btab = nullptr;      // Ed25519 base-point comb (16 bits: the one-lane kernel's), built on first use
    sbv::aniels* d_ed_bcomb = nullptr;  // the grouped step's wider comb of B (SBV_ED_B_BITS, default 20: 13 x 2^19 entries = 654 MB); == d_btab at 16 bits
    int ed_bbits = 16;
    u32 ed_bpitch = 96;                 // bytes from one entry of d_ed_bcomb to the next (ed25519_core.h: edcomb)
    sbv::kapt* d_k256_gtab = nullptr;   // secp256k1 comb of G (17 x 32768 entries), built on first use
    sbv::kapt* d_k256_gcomb = nullptr;  // the grouped step's wider comb of G (SBV_K256_G_BITS, default 20: 13 x 2^19 entries), built on first use
    int k256_gbits = 16;
    // latency form of small registered-key batches (k_p256_verify_prepared_small): page-locked buffers mapped into the device's
    // address space — input records + slots, one verdict byte per signature + the completion counter the host polls
    uint8_t* h_small_in = nullptr; void* d_small_in = nullptr;
    uint8_t* h_small_out = nullptr; void* d_small_out = nullptr;
    bool small_enabled = true;
    // registered keys
    sbv::apt* d_ktab = nullptr;
    uint8_t* d_kvalid = nullptr;
    u32* d_slots = nullptr;             // staging for the host-pointer keyed entry (cap entries)
    size_t key_cap = 0, nkeys = 0;
    std::unordered_map<std::string, u32> key_index;
    // wide combs of the slots sbv_p256_widen_keys named (p256_comb29.h: widekeys): comb w of gcomb_entries(kwide_bits) entries belongs
    // to slot wide_slots[w]; d_kwidx[slot] = w or SBV_WIDE_NONE (key_cap entries, grown with d_kvalid)
    sbv::apt* d_kwide = nullptr;
    u32* d_kwidx = nullptr;
    size_t kwide_cap = 0;
    std::vector<u32> wide_slots;
    int kwide_bits = 20;
    bool wide_nofit = false;            // the last widen_slots() found no room for the pool: sync_registry does not retry until the widened list changes
    size_t wide_nofit_at = (size_t)-1;
    bool kwide_auto = true;             // width by the number of widened keys: 20 bits up to kWideAutoSplit keys, 16 beyond (sbv_p256_wide_keys)
    u32 kwide_max = 64;
    int profiling = 0;                     // 0 off, 1 = step triples + dominant-kernel pairs, 2 = dominant-kernel pairs only
    std::vector<hipEvent_t> prof_events;   // triples: before prep, after prep, after verify
    std::vector<hipEvent_t> prof_dom;      // pairs around the dominant kernel of grouped batches (nullptr pair = ungrouped)
    size_t prof_dom_used = 0;
    size_t prof_used = 0;
};

// One Context per device.  A replica process of the reference injects ONE Verifier (pkg/consensus/consensus.go:35, 107) and
// that process drives every GPU of the node: sbv_init(d) creates device d's context (the first one becomes the default that
// the single-device entry points use), sbv_init_all() creates one per visible gfx950 device, and the sharded / _on entries
// address them.  Contexts are never freed (a thread may still hold a pointer while another shuts down): sbv_shutdown tears
// the device resources down and marks them not ready.
constexpr int kMaxDevices = 16;
// Process-wide settings (sbv_p256_set_grouping / sbv_p256_key_cache / sbv_profile_enable): remembered here, copied into every
// context when it is initialised and applied to every live context when they change — a setter called before sbv_init is not
// lost, and after sbv_init_all it configures ALL devices, not just the default one.  Guarded by g_set_mu (a leaf lock).
struct Settings {
    bool group_enabled = true; size_t group_min_batch = 64, group_min_batch_cold = (size_t)1 << 17, group_min_batch_ed = (size_t)1 << 18, group_min_batch_k256 = (size_t)1 << 17; u32 group_min_count = 0, group_max = 65536;
    bool kc_on[3] = {true, true, true}; u32 kc_caps[3] = {16384, 1024, 1024};
    int profiling = 0;
    int wide_bits = SBV_WIDE_BITS_AUTO; u32 wide_max = 64;          // sbv_p256_wide_keys; env SBV_KEYED_WIDE_BITS (0 = off, 1 = auto), SBV_KEYED_WIDE_MAX
    u32 hot_keys = 1024, hot_min_hits = 4096;                       // sbv_p256_hot_keys; env SBV_HOT_KEYS (0 = off), SBV_HOT_MIN_HITS
    u32 ed_hot_keys = 1024, ed_hot_min_hits = 4096;                 // sbv_ed25519_hot_keys; env SBV_ED_HOT_KEYS (0 = off), SBV_ED_HOT_MIN_HITS
} g_settings;
std::mutex g_set_mu;
std::unique_ptr<Context> g_ctxs[kMaxDevices];
Context g_null;                      // stand-in before sbv_init: ready == false
Context* g_def = nullptr;
std::mutex g_mu;                     // the registry above, init / shutdown, and the RCCL communicators
Context* default_ctx() {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_def ? g_def : &g_null;
}
// The key registry is PROCESS-wide (round 5): types.Signature.ID selects a consenter's key on whichever device a shard of a
// decision batch lands (internal/bft/viewchanger.go:681-727; BASELINE configs[3]: the commit signatures of 50 000 proposals over the
// 8 GPUs of a node), so a slot must mean the same key on every device.  g_reg is the one source of slot numbers; every context
// holds a replica of the combs (8-bit comb per key, wide combs of the widened slots) that sync_registry() brings up to date —
// eagerly when a key is registered / widened and when a device is initialised later, and once more in front of every sharded
// registered-key call (a device whose replication failed is retried there and fails THAT call, never silently rejects).
// Lock order: g_reg_mu -> g_mu -> Context::mu.  Registration takes g_reg_mu exclusively, the sharded registered-key entry shared.
struct Registry {
    std::vector<std::string> keys;                    // slot -> the 64 key bytes
    std::unordered_map<std::string, u32> index;       // key bytes -> slot
    std::vector<u32> wide;                            // slots named to sbv_p256_widen_keys, in the order they were named
} g_reg;
std::shared_mutex g_reg_mu;
// every initialised context, for the process-wide setters
std::vector<Context*> live_contexts() {
    std::lock_guard<std::mutex> lk(g_mu);
    std::vector<Context*> v;
    for (auto& up : g_ctxs) if (up) v.push_back(up.get());
    return v;
}
// the HIP device behind context `idx` (g_mu held by the caller, or the context known to be alive)
int hipdev_of(int idx) { return idx >= 0 && idx < kMaxDevices && g_ctxs[idx] ? g_ctxs[idx]->hip_dev : idx; }
// every single-device entry point: lock the default context for the duration of the call
#define SBV_ENTER(c)                                  \
    Context* cp_ = default_ctx();                     \
    std::lock_guard<std::mutex> lk(cp_->mu);          \
    Context& c = *cp_
// Error text of the calling thread's last failing call: thread_local, so sbv_last_error() never races with another
// thread's failure (the host Verifier calls it from many threads at once exactly when the device is faulting).
thread_local std::string g_err;

int fail(int code, const char* what, hipError_t e) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return code;
}
#define HIP_TRY(code, call)                                     \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return fail(code, #call, e_);     \
    } while (0)

// Fill device memory and WAIT for the fill.  hipMemset of device memory returns before the fill has run — it is ordered in the NULL
// stream only — and every stream of this library is created hipStreamNonBlocking: nothing orders their kernels behind the null stream.
// Rounds 4-5 initialised the key-table cache (hash table, entry counter), the class bytes and the hot-key bookkeeping with bare
// hipMemset calls and launched the first grouped batch right behind them; normally the fill wins by milliseconds.  With eight contexts
// on one GPU (SBV_LOGICAL_DEVICES, round 6) the null stream is shared and busy with the other contexts' table uploads: in one start-up of
// ~60 a context's late fill wiped the cache entries its first batch had just published, the next batch re-inserted every key into
// re-used slots, and the wide combs promoted for the old slot numbers then served other keys — honest signatures rejected from the third
// batch on (tools/stress_logical.py; profiles/r06/stress_logical_*).  A device-side race of this kind is a correctness bug on any
// device count.
hipError_t memset_now(void* p, int v, size_t bytes) {
    hipError_t e = hipMemset(p, v, bytes);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}

constexpr size_t kMaxChunk = (size_t)1 << 21;   // tuples per launch; bounds scratch at ~3.3 GB
// Groups per batch of the Ed25519 / secp256k1 grouped steps.  Their tables are always full 8-bit combs (no rows-only class) at the
// threshold of 64 uses they always had, so round 5's 65 536 groups of the P-256 step would only size their pools (384 KiB per key);
// they keep the capacity they were measured with.
constexpr size_t kVariantGroups = 2048;

void free_buffers(Context& c) {
    if (c.d_tuples) (void)hipFree(c.d_tuples);
    if (c.d_scratch) (void)hipFree(c.d_scratch);
    if (c.d_qtab) (void)hipFree(c.d_qtab);
    if (c.d_bitmap) (void)hipFree(c.d_bitmap);
    if (c.h_bitmap) (void)hipHostFree(c.h_bitmap);
    if (c.d_slots) (void)hipFree(c.d_slots);
    c.d_slots = nullptr;
    if (c.d_rerun) (void)hipFree(c.d_rerun);
    c.d_rerun = nullptr;
    c.d_tuples = c.d_scratch = c.d_bitmap = c.h_bitmap = nullptr;
    c.d_qtab = nullptr;
    c.cap = 0;
}

// make room for `n` tuples per launch (n <= kMaxChunk)
int ensure_capacity(Context& c, size_t n) {
    size_t want = (n + 1023) & ~(size_t)1023;
    if (want <= c.cap) return SBV_OK;
    if (c.busy_valid) { HIP_TRY(SBV_EDEVICE, hipEventSynchronize(c.busy)); c.busy_valid = false; }
    free_buffers(c);
    HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_tuples, want * SBV_TUPLE_BYTES));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_scratch, want * (6 * 32 + 1)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_qtab, want * (size_t)SBV_QTAB29_WORDS * sizeof(u32)));      // per-lane strip of the generic kernel (Ed25519: 8 x 128 B fits)
    HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_bitmap, want / 8));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_slots, want * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_rerun, want / 64));
    HIP_TRY(SBV_ENOMEM, hipHostMalloc(&c.h_bitmap, want / 8, hipHostMallocDefault));
    c.cap = want;
    return SBV_OK;
}

sbv::Scratch scratch_view(const Context& c) {
    sbv::Scratch s;
    u32* base = reinterpret_cast<u32*>(c.d_scratch);
    const size_t stride = c.cap * 8;
    s.r = base;
    s.u1 = base + stride;
    s.u2 = base + 2 * stride;
    s.qx = base + 3 * stride;
    s.qy = base + 4 * stride;
    s.sm = base + 5 * stride;
    s.ok = c.d_scratch + 6 * stride * sizeof(u32);
    s.cap = c.cap;
    return s;
}

std::vector<hipEvent_t*> group_events(Context& c) {
    sbv::GroupSync& y = c.gsync;
    std::vector<hipEvent_t*> v = {&y.ev_fork, &y.ev_assign, &y.ev_split, &y.ev_generic, &y.ev_cache, &y.ev_class, &y.ev_narrow, &y.ev_promote, &y.ev_promoted, &y.ev_wide};
    for (int i = 0; i < SBV_GROUP_MAX_CHUNKS; ++i) v.push_back(&y.ev_bases[i]);
    for (int i = 0; i < SBV_GROUP_MAX_CHUNKS; ++i) v.push_back(&y.ev_tables[i]);
    return v;
}

// Keys of the grouping / key-cache hashes (p256_group.h: hash-flooding defence): one fresh 32-bit value per table from the OS's
// entropy source, never exported.  SBV_HASH_SEED=<hex> pins it — for tests that must reproduce a collision pattern (seed 0 is the
// unkeyed hash of rounds 1-4), never for production.
u32 fresh_hash_seed() {
    static const char* pinned = getenv("SBV_HASH_SEED");
    if (pinned) return (u32)strtoul(pinned, nullptr, 16);
    static std::mutex mu;
    static std::random_device rd;
    std::lock_guard<std::mutex> lk(mu);
    const u32 t = (u32)std::chrono::steady_clock::now().time_since_epoch().count();
    return (u32)rd() ^ ((u32)rd() << 16) ^ (t * 0x9E3779B1u);
}

// device arrays of one persistent key-table cache (p256_group.h: KeyCache) for `K` keys
int key_cache_alloc(sbv::KeyCache& kc, size_t K, bool enabled) {
    size_t kht = 1024;
    while (kht < 4 * (K ? K : 1)) kht *= 2;
    HIP_TRY(SBV_ENOMEM, hipMalloc(&kc.ht, kht * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&kc.keys, (K ? K : 1) * 16 * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&kc.count, 4 * sizeof(u32)));
    HIP_TRY(SBV_EDEVICE, memset_now(kc.ht, 0, kht * sizeof(u32)));
    HIP_TRY(SBV_EDEVICE, memset_now(kc.count, 0, 4 * sizeof(u32)));
    kc.ht_mask = (u32)(kht - 1);
    kc.cap = (u32)K;
    kc.enabled = enabled ? 1u : 0u;
    kc.seed = fresh_hash_seed();           // fixed for as long as this table holds entries
    return SBV_OK;
}
void key_cache_free(sbv::KeyCache& kc) {
    void* ptrs[] = {kc.ht, kc.keys, kc.count};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    kc = sbv::KeyCache{};
}
// forget every cached key (the tables stay where they are; nothing points at them any more)
hipError_t key_cache_forget(sbv::KeyCache& kc) {
    if (!kc.ht) return hipSuccess;
    hipError_t e = memset_now(kc.ht, 0, ((size_t)kc.ht_mask + 1) * sizeof(u32));
    if (e == hipSuccess) e = memset_now(kc.count, 0, 4 * sizeof(u32));
    return e;
}

// the P-256 cache was emptied: its slots' wide combs belong to nobody any more
hipError_t hot_forget(sbv::GroupBuffers& b) {
    if (!b.kwide) return hipSuccess;
    hipError_t e = memset_now(b.kwide, 0xFF, (size_t)b.kc.cap * sizeof(u32));
    if (e == hipSuccess) e = memset_now(b.khits, 0, (size_t)b.kc.cap * sizeof(u32));
    if (e == hipSuccess) e = memset_now(b.hot, 0, 4 * sizeof(u32));
    if (e == hipSuccess && b.wowner) e = memset_now(b.wowner, 0xFF, (size_t)b.wide_cap * sizeof(u32));
    return e;
}

// the same for the Ed25519 scheme's pool (ed25519_group.h: hot keys)
hipError_t ed_hot_forget(sbv::EdGroupBuffers& e) {
    if (!e.kwide) return hipSuccess;
    hipError_t r = memset_now(e.kwide, 0xFF, (size_t)e.kc.cap * sizeof(u32));
    if (r == hipSuccess) r = memset_now(e.khits, 0, (size_t)e.kc.cap * sizeof(u32));
    if (r == hipSuccess) r = memset_now(e.hot, 0, 4 * sizeof(u32));
    if (r == hipSuccess) r = memset_now(e.wowner, 0xFF, (size_t)e.wide_cap * sizeof(u32));
    return r;
}
void ed_hot_free(sbv::EdGroupBuffers& e) {
    void* part[] = {e.wtab, e.kwide, e.khits, e.hot, e.plist, e.wowner, e.elist, e.ptmp, e.wide};
    for (void* q : part) if (q) (void)hipFree(q);
    e.wtab = nullptr; e.kwide = nullptr; e.khits = nullptr; e.hot = nullptr; e.plist = nullptr; e.wowner = nullptr; e.elist = nullptr;
    e.ptmp = nullptr; e.wide = nullptr; e.wide_cap = 0;
}

// keep_pools: the comb pools and key-table caches of the three schemes depend on (cache capacity, max_groups) only — a batch larger
// than any before regrows the per-tuple arrays and must leave every cached comb where it is
void free_group_buffers(Context& c, bool keep_pools = false) {
    sbv::GroupBuffers& b = c.grp;
    void* ptrs[] = {b.ht, b.rep, b.cnt, b.slot_of, b.group_rep, b.counters, b.grp_idx, b.ung_idx, b.slots, b.jbases, b.bases, b.jstate, b.tmp, b.acc, b.gacc,
                    b.gcount, b.grp_of, b.ung_cand, b.rec, b.tslot, b.cold, b.full, b.needfill, b.wide};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    sbv::apt* const ktab = b.ktab;
    sbv::apt* const ntab = b.ntab;
    uint8_t* const kvalid = b.kvalid;
    uint8_t* const kfull = b.kfull;
    const sbv::KeyCache kc = b.kc;
    const sbv::GroupBuffers old = b;                 // the hot-key pool travels with the other pools
    b = sbv::GroupBuffers();
    if (keep_pools) {
        b.ktab = ktab; b.ntab = ntab; b.kvalid = kvalid; b.kfull = kfull; b.kc = kc;
        b.wtab = old.wtab; b.kwide = old.kwide; b.khits = old.khits; b.hot = old.hot; b.plist = old.plist; b.pbases = old.pbases; b.ptmp = old.ptmp;
        b.wowner = old.wowner; b.elist = old.elist; b.hot_tick = old.hot_tick;
        b.wide_cap = old.wide_cap; b.promote_min = old.promote_min;
        return;
    }
    void* pool[] = {ktab, ntab, kvalid, kfull, kc.ht, kc.keys, kc.count, old.wtab, old.kwide, old.khits, old.hot, old.plist, old.pbases, old.ptmp, old.wowner, old.elist};
    for (void* p : pool) if (p) (void)hipFree(p);
    if (c.edgrp.ktab) (void)hipFree(c.edgrp.ktab);
    if (c.edgrp.okb) (void)hipFree(c.edgrp.okb);
    if (c.edgrp.ungxy) (void)hipFree(c.edgrp.ungxy);
    if (c.edgrp.kvalid) (void)hipFree(c.edgrp.kvalid);
    key_cache_free(c.edgrp.kc);
    ed_hot_free(c.edgrp);
    c.edgrp = sbv::EdGroupBuffers();
    if (c.k256pool.ktab) (void)hipFree(c.k256pool.ktab);
    if (c.k256pool.kvalid) (void)hipFree(c.k256pool.kvalid);
    key_cache_free(c.k256pool.kc);
    c.k256pool = sbv::KeyPool();
}

bool wide_pool_fits(const Context& c, size_t extra_bytes, unsigned percent = 15);
// HBM that everything sized by (cache capacity K, groups per batch G) takes: the comb pool (270 KiB per slot), the compact rows
// (33 KiB), the class bytes, and per group the builder's base records and chain state.
size_t group_pool_bytes(size_t K, size_t G) {
    const size_t slots = K + G;
    return slots * ((size_t)SBV_KEYTAB_ENTRIES * sizeof(sbv::apt) + (size_t)(SBV_GTAB_WINDOWS * 16) * sizeof(sbv::apt) + 2) +
           G * ((size_t)SBV_GTAB_WINDOWS * (8 * 36) * sizeof(u32) + 36 * sizeof(u32) + 64);
}
// The pools are sized for the workload (65 536 groups per batch + 16 384 cached keys = 25 GB: nothing on a 288 GB device), but a
// device that cannot hold them must still VERIFY (round 5 failed every grouped batch with SBV_ENOMEM below ~30 GB free — a
// liveness failure for a BFT verifier): halve the groups per batch, then the cache, until the pools fit into a quarter of the device
// (of this context's share of it, SBV_LOGICAL_DEVICES) and leave `keep_free` for the per-tuple arrays.  Smaller pools change rates
// (fewer keys get tables, the rest take the one-lane kernel), never verdicts.  Returns false when even the smallest pools do not fit.
bool fit_group_pools(const Context& c, size_t& K, size_t& G) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return true;            // no figure: try as asked, the allocation decides
    const size_t share = (size_t)(c.mem_share > 0 ? c.mem_share : 1);
    const size_t keep_free = (size_t)6 << 30;
    size_t budget = total_b / 4 / share;
    const size_t avail = free_b > keep_free ? free_b - keep_free : 0;
    if (budget > avail) budget = avail;
    if (const char* e = getenv("SBV_POOL_BUDGET_MB")) { const long v = atol(e); if (v > 0) budget = (size_t)v << 20; }     // tests: a "small device"
    while (group_pool_bytes(K, G) > budget) {
        if (G > 1024 && G >= K) G /= 2;
        else if (K > 64) K /= 2;
        else if (G > 64) G /= 2;
        else return false;
    }
    return true;
}
int ensure_group_buffers(Context& c, size_t n) {
    sbv::GroupBuffers& b = c.grp;
    if (b.cap >= n && b.max_groups == c.group_max && b.gacc_cap == c.cap && b.kc.cap == c.kc_caps[0]) {
        b.min_count = c.group_min_count ? c.group_min_count : SBV_GROUP_MIN_COUNT_DEFAULT;
        b.sample_shift = c.group_min_count ? -1 : c.group_sample_shift;
        b.full_min = c.group_full_min;
        b.promote_min = c.hot_min_hits;
        b.kc.enabled = c.kc_on[0] ? 1u : 0u;
        return SBV_OK;
    }
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    const bool keep_pools = b.ktab && b.ntab && b.kvalid && b.kfull && b.kc.ht && b.max_groups == c.group_max && b.kc.cap == c.kc_caps[0];
    free_group_buffers(c, keep_pools);
    const size_t cap = (n + 1023) & ~(size_t)1023;
    size_t ht = 1024;
    while (ht < 2 * cap) ht *= 2;
    if (!keep_pools) {
        size_t Kf = c.kc_caps[0], Gf = c.group_max;
        if (!fit_group_pools(c, Kf, Gf)) { g_err = "grouping pools do not fit into this device's free memory"; return SBV_ENOMEM; }
        if (Kf != c.kc_caps[0] || Gf != c.group_max) { c.pools_shrunk = true; c.kc_caps[0] = (u32)Kf; c.group_max = (u32)Gf; }
    }
    const size_t G = c.group_max;
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.ht, ht * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.rep, cap * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.cnt, cap * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.slot_of, cap * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.group_rep, G * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.counters, SBV_GROUP_COUNTERS * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.grp_idx, cap * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.ung_idx, cap * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.slots, cap * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.gcount, 2 * G * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.grp_of, cap * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.ung_cand, cap * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.rec, c.cap * (size_t)SBV_REC_WORDS * sizeof(u32)));    // indexed like the scratch planes
    const size_t Gv = G < kVariantGroups ? G : kVariantGroups;            // what the Ed25519 / secp256k1 steps use of these arrays (variant_groups())
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.jbases, Gv * SBV_GTAB_WINDOWS * (size_t)40 * sizeof(u32)));   // Ed25519 only: 40 dwords = one Jacobian base (p256_group.h)
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.full, G));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.needfill, G));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.bases, G * SBV_GTAB_WINDOWS * (size_t)(8 * 36) * sizeof(u32)));       // p256_keytab29.h: SBV_KT29_POINTS_PER_WINDOW records of SBV_KT29_REC_WORDS
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.jstate, G * (size_t)36 * sizeof(u32)));                                  // SBV_KT29_STATE_WORDS
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.gacc, 40 * c.cap * sizeof(u32)));       // limb-major with the scratch's stride; 36 words (P-256: XYZZ, 9-limb coordinates) or 40 (Ed25519: extended, 10-limb coordinates) per tuple
    // comb pool: slots [0, kc_cap) belong to the persistent key-table cache, [kc_cap, kc_cap + G) are rebuilt per batch
    const size_t K = c.kc_caps[0];
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.tslot, G * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.cold, G));
    if (!keep_pools) {
        HIP_TRY(SBV_ENOMEM, hipMalloc(&b.ktab, (K + G) * (size_t)SBV_KEYTAB_ENTRIES * sizeof(sbv::apt)));
        HIP_TRY(SBV_ENOMEM, hipMalloc(&b.ntab, (K + G) * (size_t)(SBV_GTAB_WINDOWS * 16) * sizeof(sbv::apt)));      // compact rows: 33 KiB per slot
        HIP_TRY(SBV_ENOMEM, hipMalloc(&b.kvalid, K + G));
        HIP_TRY(SBV_ENOMEM, hipMalloc(&b.kfull, K + G));
        HIP_TRY(SBV_EDEVICE, memset_now(b.kfull, 0, K + G));
        const int krc = key_cache_alloc(b.kc, K, c.kc_on[0]);
        if (krc != SBV_OK) return krc;
        // hot keys: the pool of wide combs for promoted cache slots (p256_group.h), as large as asked for if that leaves the reserve free
        // (wide_pool_fits: 1024 keys x 35.7 MB = 36.5 GB of the 288), smaller or absent otherwise — verdicts never depend on it
        b.wide_cap = 0;
        if (c.hot_keys && K) {
            const size_t per = sbv::gcomb_entries(SBV_HOT_BITS) * sizeof(sbv::apt);
            size_t want = c.hot_keys > 4096 ? 4096 : c.hot_keys;        // k_promote_evict's bitmap of the combs handed out in a batch covers 4096
            while (want && !wide_pool_fits(c, want * per)) want /= 2;
            if (want) {
                // an OPTIONAL pool: a failed allocation leaves the hot keys off (same verdicts), it never fails the batch
                const size_t W = (257 + SBV_HOT_BITS - 1) / SBV_HOT_BITS;
                const bool got = hipMalloc(&b.wtab, want * per) == hipSuccess && hipMalloc(&b.kwide, K * sizeof(u32)) == hipSuccess &&
                                 hipMalloc(&b.khits, K * sizeof(u32)) == hipSuccess && hipMalloc(&b.hot, 4 * sizeof(u32)) == hipSuccess &&
                                 hipMalloc(&b.plist, 2 * SBV_PROMOTE_MAX * sizeof(u32)) == hipSuccess &&
                                 hipMalloc(&b.pbases, SBV_PROMOTE_MAX * 2 * W * sizeof(sbv::apt)) == hipSuccess &&
                                 hipMalloc(&b.ptmp, sbv::widetab_tmp_words(SBV_PROMOTE_MAX, SBV_HOT_BITS) * sizeof(u32)) == hipSuccess &&
                                 hipMalloc(&b.wowner, want * sizeof(u32)) == hipSuccess && hipMalloc(&b.elist, SBV_PROMOTE_MAX * sizeof(u32)) == hipSuccess &&
                                 memset_now(b.wowner, 0xFF, want * sizeof(u32)) == hipSuccess &&
                                 memset_now(b.kwide, 0xFF, K * sizeof(u32)) == hipSuccess && memset_now(b.khits, 0, K * sizeof(u32)) == hipSuccess &&
                                 memset_now(b.hot, 0, 4 * sizeof(u32)) == hipSuccess;
                if (got) {
                    b.wide_cap = (u32)want;
                } else {
                    (void)hipGetLastError();
                    void* part[] = {b.wtab, b.kwide, b.khits, b.hot, b.plist, b.pbases, b.ptmp, b.wowner, b.elist};
                    for (void* q : part) if (q) (void)hipFree(q);
                    b.wtab = nullptr; b.kwide = nullptr; b.khits = nullptr; b.hot = nullptr; b.plist = nullptr; b.pbases = nullptr; b.ptmp = nullptr;
                    b.wowner = nullptr; b.elist = nullptr;
                }
            }
        }
    }
    b.promote_min = c.hot_min_hits;
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.wide, G));
    b.kc.enabled = c.kc_on[0] ? 1u : 0u;
    {   // scratch of the table kernels: P-256 indexes it by resident lane (two table streams x SBV_TABLE_GRID_BLOCKS x 64 lanes x 675 words,
        // p256_group_kernels.hip), the Ed25519 step by (key, window): 128 x 40 raw limbs each, for the groups that step may hold
        const size_t p256_words = 2 * (size_t)4096 * 64 * (15 * 45);
        const size_t ed_words = Gv * SBV_GTAB_WINDOWS * (size_t)(SBV_GTAB_PER_WINDOW * 40);
        HIP_TRY(SBV_ENOMEM, hipMalloc(&b.tmp, (p256_words > ed_words ? p256_words : ed_words) * sizeof(u32)));
    }
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.acc, cap));
    b.ht_mask = (u32)(ht - 1);
    b.seed = fresh_hash_seed();            // the per-batch grouping table is empty at the start of every batch: any seed will do, a secret one is the point
    b.max_groups = (u32)G;
    b.min_count = c.group_min_count ? c.group_min_count : SBV_GROUP_MIN_COUNT_DEFAULT;
    b.sample_shift = c.group_min_count ? -1 : c.group_sample_shift;
    b.full_min = c.group_full_min;
    b.cap = cap;
    b.gacc_cap = c.cap;
    return SBV_OK;
}

// The Ed25519 / secp256k1 steps see the shared grouping arrays with THEIR group capacity and threshold (kVariantGroups; 64 uses unless
// sbv_p256_set_grouping named one): a view of c.grp, never stored.
u32 variant_groups(const Context& c) { return c.grp.max_groups < kVariantGroups ? c.grp.max_groups : (u32)kVariantGroups; }
sbv::GroupBuffers variant_view(const Context& c, size_t n) {
    sbv::GroupBuffers bv = c.grp;
    bv.max_groups = variant_groups(c);
    bv.min_count = c.group_min_count ? c.group_min_count : 64u;
    bv.sample_shift = -1;
    if (n < ((size_t)1 << 18) && bv.min_count > 32) bv.min_count = 32;     // no stragglers on the one-lane path below 2^18 (enqueue() has the numbers)
    return bv;
}

int ensure_ed_group_buffers(Context& c, size_t n) {
    int rc = ensure_group_buffers(c, n);
    if (rc != SBV_OK) return rc;
    sbv::EdGroupBuffers& e = c.edgrp;
    const size_t K = c.kc_caps[2];
    // The comb pool and its cache depend on (K, max_groups) only: a batch larger than any before must not empty the cache
    // (found on the GPU in round 4: the warm batch of the cache test was 384 tuples longer than the cold one and missed every key).
    const u32 vg = variant_groups(c);
    const bool pool_ok = e.ktab && e.max_groups == vg && e.kc.ht && e.kc.cap == K;
    e.promote_min = c.ed_hot_min_hits;
    if (pool_ok && e.okb && e.cap >= c.grp.cap) {
        e.kc.enabled = c.kc_on[2] ? 1u : 0u;
        return SBV_OK;
    }
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    if (e.okb) (void)hipFree(e.okb);
    if (e.ungxy) (void)hipFree(e.ungxy);
    e.okb = nullptr; e.ungxy = nullptr; e.cap = 0;
    if (!pool_ok) {
        if (e.ktab) (void)hipFree(e.ktab);
        if (e.kvalid) (void)hipFree(e.kvalid);
        key_cache_free(e.kc);
        ed_hot_free(e);
        e = sbv::EdGroupBuffers();
        // comb pool of this scheme: slots [0, K) = its persistent key-table cache, [K, K + max_groups) per batch
        HIP_TRY(SBV_ENOMEM, hipMalloc(&e.ktab, (K + vg) * (size_t)SBV_ED_KEYTAB_ENTRIES_PER_KEY * sizeof(sbv::aniels)));
        HIP_TRY(SBV_ENOMEM, hipMalloc(&e.kvalid, K + vg));
        rc = key_cache_alloc(e.kc, K, c.kc_on[2]);
        if (rc != SBV_OK) return rc;
        e.max_groups = vg;
        // hot keys of this scheme: an OPTIONAL pool of 64 MiB combs, as large as asked for if the device has the room (wide_pool_fits),
        // smaller or absent otherwise — a failed allocation leaves the feature off, it never fails the batch; verdicts never depend on it
        if (c.ed_hot_keys && K) {
            size_t want = c.ed_hot_keys > 4096 ? 4096 : c.ed_hot_keys;
            const size_t scratch = (size_t)SBV_ED_HOT_BUILD_BLOCKS * 64 * SBV_ED_HOT_TMP_WORDS * sizeof(u32);
            while (want && !wide_pool_fits(c, want * SBV_ED_HOT_COMB_BYTES + scratch, 25)) want /= 2;       // an opt-in pool: 1024 combs are 64 GiB of the 288
            if (want) {
                const bool got = hipMalloc(&e.wtab, want * SBV_ED_HOT_COMB_BYTES) == hipSuccess && hipMalloc(&e.kwide, K * sizeof(u32)) == hipSuccess &&
                                 hipMalloc(&e.khits, K * sizeof(u32)) == hipSuccess && hipMalloc(&e.hot, 4 * sizeof(u32)) == hipSuccess &&
                                 hipMalloc(&e.plist, 2 * SBV_PROMOTE_MAX * sizeof(u32)) == hipSuccess && hipMalloc(&e.ptmp, scratch) == hipSuccess &&
                                 hipMalloc(&e.wowner, want * sizeof(u32)) == hipSuccess && hipMalloc(&e.elist, SBV_PROMOTE_MAX * sizeof(u32)) == hipSuccess &&
                                 hipMalloc(&e.wide, vg) == hipSuccess;
                if (got) e.wide_cap = (u32)want;
                if (!got || ed_hot_forget(e) != hipSuccess || memset_now(e.wide, 0, vg) != hipSuccess) {
                    (void)hipGetLastError();
                    ed_hot_free(e);
                }
            }
        }
    }
    HIP_TRY(SBV_ENOMEM, hipMalloc(&e.okb, c.grp.cap));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&e.ungxy, c.grp.cap * (size_t)SBV_ED_UNGXY_WORDS * sizeof(u32)));
    e.promote_min = c.ed_hot_min_hits;
    e.cap = c.grp.cap;
    e.kc.enabled = c.kc_on[2] ? 1u : 0u;
    return SBV_OK;
}

// secp256k1: the grouping arrays of the P-256 step + this curve's own comb pool and key-table cache
int ensure_k256_group_buffers(Context& c, size_t n) {
    int rc = ensure_group_buffers(c, n);
    if (rc != SBV_OK) return rc;
    sbv::KeyPool& kp = c.k256pool;
    const size_t K = c.kc_caps[1];
    const u32 vg = variant_groups(c);
    if (kp.ktab && kp.max_groups == vg && kp.kc.cap == K) {
        kp.kc.enabled = c.kc_on[1] ? 1u : 0u;
        return SBV_OK;
    }
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    if (kp.ktab) (void)hipFree(kp.ktab);
    if (kp.kvalid) (void)hipFree(kp.kvalid);
    key_cache_free(kp.kc);
    kp = sbv::KeyPool();
    HIP_TRY(SBV_ENOMEM, hipMalloc(&kp.ktab, (K + vg) * (size_t)SBV_KEYTAB_ENTRIES * sizeof(sbv::apt)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&kp.kvalid, K + vg));
    rc = key_cache_alloc(kp.kc, K, c.kc_on[1]);
    if (rc != SBV_OK) return rc;
    kp.max_groups = vg;
    return SBV_OK;
}

// A grouped batch needs its pools; a device that cannot give them (even after fit_group_pools halved them) must still verify: the
// batch — and the next 64 grouped ones, so that a tight device does not pay a failed allocation and a device-wide synchronisation per
// call — takes the one-lane kernel instead.  false = a real error (g_last_rc); grouped is cleared on the fallback.
thread_local int g_last_rc = SBV_OK;
template <class F>
bool group_buffers_or_fallback(Context& c, size_t, F&& ensure, bool& grouped) {
    if (c.group_nomem_skip) { --c.group_nomem_skip; grouped = false; return true; }
    const int rc = ensure();
    if (rc == SBV_OK) return true;
    if (rc != SBV_ENOMEM) { g_last_rc = rc; return false; }
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();
    free_group_buffers(c);                   // whatever the failed attempt left allocated goes back to the device
    c.group_nomem_skip = 64;
    ++c.group_nomem_events;
    grouped = false;
    g_err.clear();
    return true;
}

int ensure_ed_bcomb(Context& c);
// one chunk (n <= cap) of Ed25519 tuples on `stream`: grouped step or the one-lane kernel
int enqueue_ed25519(Context& c, const uint8_t* d_tuples, size_t n, uint8_t* d_bitmap, hipStream_t stream, hipEvent_t* dom = nullptr, int* dom_pairs = nullptr) {
    // with the scheme's key-table cache on, nearly every batch takes the grouped step (as for P-256: cached keys are grouped whatever
    // their count, and a cold batch leaves its combs behind); with it off the cold crossover applies
    bool grouped = c.group_enabled && n >= (c.kc_on[2] ? c.group_min_batch : c.group_min_batch_ed);
    if (grouped && !group_buffers_or_fallback(c, n, [&] { const int r = ensure_ed_group_buffers(c, n); return r != SBV_OK ? r : ensure_ed_bcomb(c); }, grouped)) return g_last_rc;
    if (grouped) {
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.gsync.ev_fork, stream));
        if (c.edgrp.wtab && c.edgrp.kc.enabled) ++c.edgrp.hot_tick;     // the clock of the hot keys' decay
        const hipError_t ge = sbv::launch_ed25519_verify_grouped(d_tuples, n, variant_view(c, n), c.edgrp, c.d_qtab, c.d_btab, sbv::edcomb_make(c.d_ed_bcomb, c.ed_bbits, c.ed_bpitch), d_bitmap, stream, c.gsync, dom, dom_pairs);
        if (ge != hipSuccess) {          // a slot is published before its tables are built (see enqueue()): forget the cache
            (void)hipDeviceSynchronize();
            (void)key_cache_forget(c.edgrp.kc);
            (void)ed_hot_forget(c.edgrp);
            return fail(SBV_EDEVICE, "launch_ed25519_verify_grouped", ge);
        }
        return SBV_OK;
    }
    HIP_TRY(SBV_EDEVICE, sbv::launch_ed25519_verify(d_tuples, n, c.d_qtab, c.d_btab, d_bitmap, stream));
    return SBV_OK;
}

// enqueue stage A + stage B for n <= cap tuples on `stream`
// decide_n: the batch size the grouped / one-lane decision is taken on (0 = n).  A key-affine part holds 1 / G of a batch's
// tuples but each of its keys as often as the whole batch does: it groups whenever the whole batch would.
int enqueue(Context& c, const uint8_t* d_tuples, size_t n, uint8_t* d_bitmap, hipStream_t stream,
            hipEvent_t after_prep, hipEvent_t* dom = nullptr, int* dom_pairs = nullptr, bool* was_grouped = nullptr, size_t decide_n = 0) {
    const sbv::Scratch s = scratch_view(c);
    bool grouped = c.group_enabled && (decide_n ? decide_n : n) >= (c.kc_on[0] ? c.group_min_batch : c.group_min_batch_cold);
    if (grouped && !group_buffers_or_fallback(c, n, [&] { return ensure_group_buffers(c, n); }, grouped)) return g_last_rc;
    if (was_grouped) *was_grouped = grouped;
    if (grouped) {
        // Below 2^18 tuples the step is latency: one straggling tuple in the one-lane doubling kernel (2.3 ms) outlasts the
        // whole table pipeline (1.7 ms at 1024 cold keys).  A threshold of 64 uses, counted on every 8th tuple, loses 3 % of
        // the keys of a 2^17 batch over 1024 signers (128 uses each, 16 +- 3.7 samples against 8) and the step went from 1.7
        // to 2.2-2.5 ms (profiles/r03/sweep_sizes_r03a.jsonl: 2582 tuples on the one-lane path); 32 uses, counted on every
        // 4th tuple (32 +- 4.9 samples against 8), loses none.  At full size the configured threshold stands: twice the
        // counting atomics in k_group_insert, which is on the path to the G phase.
        if (n < ((size_t)1 << 18) && c.grp.min_count > 32) c.grp.min_count = 32;      // (explicit thresholds only: the default is 8)
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.gsync.ev_fork, stream));
    }
    if (grouped) {           // stage A is enqueued by the grouped launcher
        sbv::Scratch sg = s;
        sg.rec = c.gsync.sorted ? c.grp.rec : nullptr;      // stage A also writes the per-tuple records the key-sorted list reads
        // The second table stream is NOT a stream of its own: a process gets four hardware queues and the fifth stream shares
        // one with the first — here the caller's, so rows + fill of chunk 1 queued behind the first Q launch (2^20 cold 3.45 ->
        // 3.75 ms, profiles/r03/timeline_r03o.txt).  When the caller brings its own stream (device-pointer entries) the
        // context's stream is idle and takes the job; the host-pointer entries run on it themselves and keep one table stream.
        sbv::GroupSync y = c.gsync;
        if (y.tstreams > 1 && stream != c.stream) y.side_t = c.stream;
        else y.tstreams = 1;
        if (c.grp.wtab && c.grp.kc.enabled) ++c.grp.hot_tick;           // the clock of the hot keys' decay (p256_group.h)
        const hipError_t ge = sbv::launch_p256_verify_grouped(d_tuples, sg, n, c.grp, c.d_qtab, c.d_gtab, sbv::gcomb_make(c.d_g16r, c.g_bits), d_bitmap, stream, y,
                                                              after_prep, dom, dom_pairs);
        if (ge != hipSuccess) {
            // k_key_cache_insert publishes a slot before its tables are built: a step that failed half-way may leave slots
            // whose combs never were.  Forget the whole cache (best effort, after draining what did get enqueued).
            (void)hipDeviceSynchronize();
            (void)key_cache_forget(c.grp.kc);
            (void)hot_forget(c.grp);
            return fail(SBV_EDEVICE, "launch_p256_verify_grouped", ge);
        }
        return SBV_OK;
    }
    HIP_TRY(SBV_EDEVICE, sbv::launch_p256_prep(d_tuples, n, s, stream));
    if (after_prep) HIP_TRY(SBV_EDEVICE, hipEventRecord(after_prep, stream));
    if (dom) HIP_TRY(SBV_EDEVICE, hipEventRecord(dom[0], stream));     // ungrouped: the dominant kernel is all of stage B
    HIP_TRY(SBV_EDEVICE, sbv::launch_p256_verify(s, n, c.d_qtab, sbv::gcomb_make(c.d_g16r, c.g_bits), d_bitmap, c.d_rerun, stream));
    if (dom) { HIP_TRY(SBV_EDEVICE, hipEventRecord(dom[1], stream)); if (dom_pairs) *dom_pairs = 1; }
    return SBV_OK;
}

sbv::widekeys wide_of(const Context& c) { return c.wide_slots.empty() ? sbv::widekeys_none() : sbv::widekeys_make(c.d_kwide, c.d_kwidx, c.kwide_bits); }

// scratch_off: first tuple slot of the scratch planes this launch may use (the sharded registered-key entries run two pieces at a
// time, each in its own half of the planes)
int enqueue_keyed(Context& c, const uint8_t* d_rsh, const u32* d_slots, size_t n, uint8_t* d_bitmap, hipStream_t stream,
                  hipEvent_t after_prep, size_t scratch_off = 0) {
    sbv::Scratch s = scratch_view(c);
    if (scratch_off) { s.r += scratch_off; s.u1 += scratch_off; s.u2 += scratch_off; s.qx += scratch_off; s.qy += scratch_off; s.sm += scratch_off; s.ok += scratch_off; }
    HIP_TRY(SBV_EDEVICE, sbv::launch_p256_prep(d_rsh, n, s, stream, true));
    if (after_prep) HIP_TRY(SBV_EDEVICE, hipEventRecord(after_prep, stream));
    HIP_TRY(SBV_EDEVICE, sbv::launch_p256_verify_keyed(s, n, d_slots, (u32)c.nkeys, c.d_ktab, c.d_kvalid, sbv::gcomb_make(c.d_g16r, c.g_bits), wide_of(c), d_bitmap, c.d_rerun, stream));
    return SBV_OK;
}

constexpr size_t kKeyTabBytes = (size_t)SBV_KEYTAB_ENTRIES * sizeof(sbv::apt);

int ensure_key_capacity(Context& c, size_t want) {
    if (want <= c.key_cap) return SBV_OK;
    size_t cap = c.key_cap ? c.key_cap : 16;
    while (cap < want) cap *= 2;
    sbv::apt* nt = nullptr;
    uint8_t* nv = nullptr;
    u32* nw = nullptr;
    HIP_TRY(SBV_ENOMEM, hipMalloc(&nt, cap * kKeyTabBytes));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&nv, cap));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&nw, cap * sizeof(u32)));
    HIP_TRY(SBV_EDEVICE, memset_now(nv, 0, cap));
    HIP_TRY(SBV_EDEVICE, memset_now(nw, 0xFF, cap * sizeof(u32)));      // SBV_WIDE_NONE
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());          // nothing in flight may still read the old tables
    if (c.nkeys) {
        HIP_TRY(SBV_EDEVICE, hipMemcpy(nt, c.d_ktab, c.nkeys * kKeyTabBytes, hipMemcpyDeviceToDevice));
        HIP_TRY(SBV_EDEVICE, hipMemcpy(nv, c.d_kvalid, c.nkeys, hipMemcpyDeviceToDevice));
        HIP_TRY(SBV_EDEVICE, hipMemcpy(nw, c.d_kwidx, c.nkeys * sizeof(u32), hipMemcpyDeviceToDevice));
        HIP_TRY(SBV_EDEVICE, hipStreamSynchronize(nullptr));     // device-to-device copies are ordered in the null stream only (see memset_now)
    }
    if (c.d_kwidx) (void)hipFree(c.d_kwidx);
    c.d_kwidx = nw;
    if (c.d_ed_bcomb && c.d_ed_bcomb != c.d_btab) (void)hipFree(c.d_ed_bcomb);
    c.d_ed_bcomb = nullptr;
    if (c.d_btab) (void)hipFree(c.d_btab);
    c.d_btab = nullptr;
    if (c.d_k256_gcomb == c.d_k256_gtab) c.d_k256_gcomb = nullptr;      // 16-bit configuration: the grouped step borrows this table
    if (c.d_k256_gtab) (void)hipFree(c.d_k256_gtab);
    c.d_k256_gtab = nullptr;
    if (c.d_msgs) (void)hipFree(c.d_msgs);
    if (c.d_sigs) (void)hipFree(c.d_sigs);
    if (c.d_moff) (void)hipFree(c.d_moff);
    if (c.d_soff) (void)hipFree(c.d_soff);
    c.d_msgs = c.d_sigs = nullptr; c.d_moff = c.d_soff = nullptr; c.msgs_cap = c.sigs_cap = c.moff_cap = c.soff_cap = 0;
    if (c.d_ktab) (void)hipFree(c.d_ktab);
    if (c.d_kvalid) (void)hipFree(c.d_kvalid);
    c.d_ktab = nt;
    c.d_kvalid = nv;
    c.key_cap = cap;
    return SBV_OK;
}

double ms_between(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.0;
    return ms;
}

}  // namespace

extern "C" int sbv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return SBV_ENODEV;
    return n;
}

namespace {
// The G combs are the same for every device: built once per process on the host, uploaded to each device.
std::vector<sbv::apt> g_h_gtab, g_h_g16r;
int g_gbits = 20;                     // window width of the carry-free kernels' comb of G (SBV_G_BITS: 12..22)
std::once_flag g_tables_once;
void build_host_tables() {
    if (const char* e = getenv("SBV_G_BITS")) { const int v = atoi(e); if (v >= 12 && v <= 22) g_gbits = v; }
    g_h_gtab.resize(SBV_G16_ENTRIES);
    sbv::host_build_g16(g_h_gtab.data());                 // 16-bit comb for the generic kernel, 35.7 MB, 17 host threads
    const size_t cnt = sbv::gcomb_entries(g_gbits);
    g_h_g16r.resize(cnt);
    if (g_gbits == 16) {
        sbv::host_convert_table_r261(g_h_gtab.data(), g_h_g16r.data(), cnt);
    } else {                                              // 20 bits: 13 x 2^19 entries = 436 MB, built once per process
        std::vector<sbv::apt> tmp(cnt);
        sbv::host_build_gcomb(g_gbits, tmp.data());
        sbv::host_convert_table_r261(tmp.data(), g_h_g16r.data(), cnt);
    }
}

// contexts sbv_init accepts / sbv_init_all creates: the visible GPUs, or SBV_LOGICAL_DEVICES of them (1..kMaxDevices)
int logical_devices(int ndev) {
    if (const char* e = getenv("SBV_LOGICAL_DEVICES")) { const int v = atoi(e); if (v >= 1) return v > kMaxDevices ? kMaxDevices : v; }
    return ndev > kMaxDevices ? kMaxDevices : ndev;
}
// c.mu held by the caller
int init_context(Context& c, int device) {
    if (c.ready) return SBV_OK;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_err = "no HIP device visible (libsbv has no CPU fallback)";
        return SBV_ENODEV;
    }
    // SBV_LOGICAL_DEVICES=G (round 6): G contexts over the visible GPUs, context i on HIP device i % ndev, every pool budget divided
    // by the contexts that share a GPU.  One physical MI355X then rehearses the whole multi-device path of the sharded entries — shard
    // offsets > 0, a host thread and two upload slots per shard, idle contexts, the host-side gather (RCCL needs one rank per
    // physical device) — and a node with fewer GPUs than a deployment's plan still runs it.
    const int logical = logical_devices(ndev);
    if (device < 0 || device >= logical) { g_err = "device index out of range"; return SBV_EINVAL; }
    const int hip_dev = device % ndev;
    c.hip_dev = hip_dev;
    c.mem_share = (logical + ndev - 1) / ndev;
    HIP_TRY(SBV_ENODEV, hipSetDevice(hip_dev));
    hipDeviceProp_t prop;
    HIP_TRY(SBV_ENODEV, hipGetDeviceProperties(&prop, hip_dev));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        g_err = std::string("device is ") + prop.gcnArchName + ", libsbv is built for gfx950 only";
        return SBV_ENODEV;
    }
    HIP_TRY(SBV_ENODEV, hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    // The side streams carry the grouping kernels and the table-building chains (few lanes, long dependent chains; the Q phase
    // waits for them).  Highest queue priority and CU masks for them were measured and rejected in round 3.
    for (hipStream_t* st : {&c.gsync.side_a, &c.gsync.side_b}) HIP_TRY(SBV_ENODEV, hipStreamCreateWithFlags(st, hipStreamNonBlocking));
    for (hipEvent_t* ev : group_events(c)) HIP_TRY(SBV_ENODEV, hipEventCreateWithFlags(ev, hipEventDisableTiming));
    {   // every knob back to its default: a context initialised again re-reads the environment
        const sbv::GroupSync d;
        c.gsync.sorted = d.sorted; c.gsync.tstreams = d.tstreams; c.gsync.gsplit_min = d.gsplit_min; c.gsync.coop_max = d.coop_max;
    }
    c.gsync.chunks = 2;
    if (const char* e = getenv("SBV_GROUP_CHUNKS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= SBV_GROUP_MAX_CHUNKS) c.gsync.chunks = v;
    }
    if (const char* e = getenv("SBV_GROUP_TSTREAMS")) { const int v = atoi(e); if (v >= 1 && v <= 2) c.gsync.tstreams = v; }
    if (const char* e = getenv("SBV_GPHASE_SPLIT_MIN")) c.gsync.gsplit_min = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("SBV_GROUP_COOP_MAX")) { const size_t v = (size_t)strtoull(e, nullptr, 10); c.gsync.coop_max = v > 32768 ? 32768 : v; }
    if (const char* e = getenv("SBV_GROUP_SORT")) c.gsync.sorted = atoi(e) != 0;
    for (auto& ev : c.ev) HIP_TRY(SBV_ENODEV, hipEventCreate(&ev));
    HIP_TRY(SBV_ENODEV, hipEventCreateWithFlags(&c.busy, hipEventDisableTiming));
    // fixed-base table: computed once on the host with the same field code, then resident in HBM
    // (16-bit comb, 35.7 MB: u1*G is 17 mixed additions; built by 17 host threads in ~0.1 s)
    const size_t gcount = SBV_G16_ENTRIES;
    std::call_once(g_tables_once, build_host_tables);
    HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_gtab, gcount * sizeof(sbv::apt)));
    HIP_TRY(SBV_EDEVICE, hipMemcpy(c.d_gtab, g_h_gtab.data(), gcount * sizeof(sbv::apt), hipMemcpyHostToDevice));
    c.g_bits = g_gbits;
    HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_g16r, g_h_g16r.size() * sizeof(sbv::apt)));
    HIP_TRY(SBV_EDEVICE, hipMemcpy(c.d_g16r, g_h_g16r.data(), g_h_g16r.size() * sizeof(sbv::apt), hipMemcpyHostToDevice));
    {
        std::lock_guard<std::mutex> lk(g_set_mu);
        c.group_enabled = g_settings.group_enabled; c.group_min_batch = g_settings.group_min_batch;
        c.group_min_batch_cold = g_settings.group_min_batch_cold; c.group_min_batch_ed = g_settings.group_min_batch_ed; c.group_min_batch_k256 = g_settings.group_min_batch_k256;
        c.group_min_count = g_settings.group_min_count; c.group_max = g_settings.group_max;
        for (int k = 0; k < 3; ++k) { c.kc_on[k] = g_settings.kc_on[k]; c.kc_caps[k] = g_settings.kc_caps[k]; }
        c.profiling = g_settings.profiling;
        c.kwide_auto = g_settings.wide_bits == SBV_WIDE_BITS_AUTO; c.kwide_bits = c.kwide_auto ? 20 : g_settings.wide_bits; c.kwide_max = g_settings.wide_max;
        c.hot_keys = g_settings.hot_keys; c.hot_min_hits = g_settings.hot_min_hits;
        c.ed_hot_keys = g_settings.ed_hot_keys; c.ed_hot_min_hits = g_settings.ed_hot_min_hits;
    }
    if (const char* e = getenv("SBV_HOT_KEYS")) { const long v = atol(e); if (v >= 0 && v <= 4096) c.hot_keys = (u32)v; }
    if (const char* e = getenv("SBV_HOT_MIN_HITS")) { const long v = atol(e); if (v >= 1) c.hot_min_hits = (u32)v; }
    if (const char* e = getenv("SBV_ED_HOT_KEYS")) { const long v = atol(e); if (v >= 0 && v <= 4096) c.ed_hot_keys = (u32)v; }
    if (const char* e = getenv("SBV_ED_HOT_MIN_HITS")) { const long v = atol(e); if (v >= 1) c.ed_hot_min_hits = (u32)v; }
    if (const char* e = getenv("SBV_KEYED_WIDE_BITS")) { const int v = atoi(e); if (v == 0) c.kwide_max = 0; else if (v == SBV_WIDE_BITS_AUTO) c.kwide_auto = true; else if (v >= 10 && v <= 20) { c.kwide_bits = v; c.kwide_auto = false; } }
    if (const char* e = getenv("SBV_KEYED_WIDE_MAX")) { const long v = atol(e); if (v >= 0 && v <= 4096) c.kwide_max = (u32)v; }
    if (const char* e = getenv("SBV_GROUP_SAMPLE_SHIFT")) { const int v = atoi(e); if (v >= 0 && v <= 6) c.group_sample_shift = v; }
    if (const char* e = getenv("SBV_FULL_TABLE_MIN")) { const long v = atol(e); if (v >= 0) c.group_full_min = (u32)v; }
    if (const char* e = getenv("SBV_SMALL")) c.small_enabled = e[0] != '0';
    if (const char* e = getenv("SBV_GROUP")) c.group_enabled = e[0] != '0';
    if (const char* e = getenv("SBV_GROUP_MIN_BATCH")) { const long v = atol(e); if (v > 0) c.group_min_batch = c.group_min_batch_cold = c.group_min_batch_ed = c.group_min_batch_k256 = (size_t)v; }
    c.device = device;
    c.ready = true;
    g_err.clear();
    return SBV_OK;
}

Context* context_of(int device, bool create) {          // g_mu held by the caller
    if (device < 0 || device >= kMaxDevices) return nullptr;
    if (!g_ctxs[device] && create) g_ctxs[device].reset(new Context());
    return g_ctxs[device].get();
}
}  // namespace

namespace { int sync_registry(Context& c); }

extern "C" int sbv_init(int device) {
    std::shared_lock<std::shared_mutex> rl(g_reg_mu);
    Context* c;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        c = context_of(device, true);
        if (!c) { g_err = "device index out of range"; return SBV_EINVAL; }
    }
    int rc;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        const bool was_ready = c->ready;
        rc = init_context(*c, device);
        // a device that joins after keys were registered gets its replica of the registry now (best effort: the sharded
        // registered-key entry retries and reports)
        if (rc == SBV_OK && !was_ready && !g_reg.keys.empty()) (void)sync_registry(*c);
    }
    if (rc == SBV_OK) {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_def) g_def = c;                          // the first initialised device is the default of the single-device entries
    }
    return rc;
}

namespace {
void rccl_teardown();
// c.mu held by the caller
// Per-device bitmap buffers of the multi-device entries.  d_gather / d_q belong to the ONE multi-shard call in flight
// (g_sharded_mu is held from its first worker to its final copy: the all-gather reads every device's d_gather after the
// workers dropped their context locks).  d_on serves the calls that stay on one device (sbv_p256_verify_batch_on and the
// replica route of the sharded entry) and is only touched under that device's context lock.
struct ShardBuffers { uint8_t* d_gather = nullptr; size_t gather_cap = 0; uint8_t* d_q = nullptr; size_t q_cap = 0;
                      uint8_t* d_on = nullptr; size_t on_cap = 0; uint8_t* d_onq = nullptr; size_t onq_cap = 0;
                      uint8_t* d_full = nullptr; size_t full_cap = 0;          // key-affine mode: the whole batch on every device
                      uint8_t* d_stage2 = nullptr; size_t stage2_cap = 0;      // second upload slot of verify_shard (the first is Context::d_tuples)
                      uint8_t* d_slots2 = nullptr; size_t slots2_cap = 0;      // ... and of verify_shard_keyed's key slots (the first is Context::d_slots)
                      uint8_t* d_msgs2 = nullptr; size_t msgs2_cap = 0; uint8_t* d_sigs2 = nullptr; size_t sigs2_cap = 0;       // verify_shard_msgs: second upload slot
                      uint8_t* d_moff2 = nullptr; size_t moff2_cap = 0; uint8_t* d_soff2 = nullptr; size_t soff2_cap = 0; };  // (the first is Context::d_msgs / d_sigs / d_moff / d_soff)
ShardBuffers g_shard[kMaxDevices];
std::mutex g_sharded_mu;
// staging of the key-affine partition (part_enqueue below), per device, under that device's context lock
struct PartBuffers { uint8_t* d_dense = nullptr; u32* d_idx = nullptr; u32* d_count = nullptr; uint8_t* d_bits = nullptr; u32* h_count = nullptr; size_t cap = 0; };
PartBuffers g_part[kMaxDevices];

int shutdown_context(Context& c) {
    if (!c.ready) return SBV_OK;
    (void)hipSetDevice(c.hip_dev);
    (void)hipDeviceSynchronize();
    free_buffers(c);
    free_group_buffers(c);
    if (c.device >= 0 && c.device < kMaxDevices) {
        ShardBuffers& sb = g_shard[c.device];
        if (sb.d_gather) (void)hipFree(sb.d_gather);
        if (sb.d_q) (void)hipFree(sb.d_q);
        if (sb.d_on) (void)hipFree(sb.d_on);
        if (sb.d_onq) (void)hipFree(sb.d_onq);
        if (sb.d_full) (void)hipFree(sb.d_full);
        if (sb.d_stage2) (void)hipFree(sb.d_stage2);
        if (sb.d_slots2) (void)hipFree(sb.d_slots2);
        for (uint8_t* p : {sb.d_msgs2, sb.d_sigs2, sb.d_moff2, sb.d_soff2}) if (p) (void)hipFree(p);
        sb = ShardBuffers();
        PartBuffers& pb = g_part[c.device];
        if (pb.d_dense) (void)hipFree(pb.d_dense);
        if (pb.d_idx) (void)hipFree(pb.d_idx);
        if (pb.d_bits) (void)hipFree(pb.d_bits);
        if (pb.d_count) (void)hipFree(pb.d_count);
        if (pb.h_count) (void)hipHostFree(pb.h_count);
        pb = PartBuffers();
    }
    if (c.d_gtab) (void)hipFree(c.d_gtab);
    c.d_gtab = nullptr;
    if (c.d_g16r) (void)hipFree(c.d_g16r);
    c.d_g16r = nullptr;
    if (c.d_ed_bcomb && c.d_ed_bcomb != c.d_btab) (void)hipFree(c.d_ed_bcomb);
    c.d_ed_bcomb = nullptr;
    if (c.d_btab) (void)hipFree(c.d_btab);
    c.d_btab = nullptr;
    if (c.d_k256_gcomb && c.d_k256_gcomb != c.d_k256_gtab) (void)hipFree(c.d_k256_gcomb);
    c.d_k256_gcomb = nullptr;
    if (c.d_k256_gtab) (void)hipFree(c.d_k256_gtab);
    c.d_k256_gtab = nullptr;
    if (c.d_msgs) (void)hipFree(c.d_msgs);
    if (c.d_sigs) (void)hipFree(c.d_sigs);
    if (c.d_moff) (void)hipFree(c.d_moff);
    if (c.d_soff) (void)hipFree(c.d_soff);
    c.d_msgs = c.d_sigs = nullptr; c.d_moff = c.d_soff = nullptr; c.msgs_cap = c.sigs_cap = c.moff_cap = c.soff_cap = 0;
    if (c.d_ktab) (void)hipFree(c.d_ktab);
    if (c.d_kvalid) (void)hipFree(c.d_kvalid);
    c.d_ktab = nullptr; c.d_kvalid = nullptr; c.key_cap = c.nkeys = 0;
    if (c.d_kwide) (void)hipFree(c.d_kwide);
    if (c.d_kwidx) (void)hipFree(c.d_kwidx);
    c.d_kwide = nullptr; c.d_kwidx = nullptr; c.kwide_cap = 0; c.wide_slots.clear();
    c.key_index.clear();
    if (c.h_small_in) (void)hipHostFree(c.h_small_in);
    if (c.h_small_out) (void)hipHostFree(c.h_small_out);
    c.h_small_in = c.h_small_out = nullptr; c.d_small_in = c.d_small_out = nullptr;
    for (auto& ev : c.ev) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    for (auto& ev : c.prof_events) (void)hipEventDestroy(ev);
    c.prof_events.clear();
    for (auto& ev : c.prof_dom) (void)hipEventDestroy(ev);
    c.prof_dom.clear();
    c.prof_dom_used = 0;
    c.prof_used = 0;
    if (c.busy) { (void)hipEventDestroy(c.busy); c.busy = nullptr; }
    if (c.stream) { (void)hipStreamDestroy(c.stream); c.stream = nullptr; }
    if (c.copy_stream) { (void)hipStreamDestroy(c.copy_stream); c.copy_stream = nullptr; }
    for (StageSlot& sl : c.stage) {
        if (sl.d_tuples) (void)hipFree(sl.d_tuples);
        if (sl.d_bitmap) (void)hipFree(sl.d_bitmap);
        if (sl.h_bitmap) (void)hipHostFree(sl.h_bitmap);
        for (auto& ev : sl.ev) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
        sl = StageSlot();
    }
    for (hipStream_t* st : {&c.gsync.side_a, &c.gsync.side_b}) if (*st) { (void)hipStreamDestroy(*st); *st = nullptr; }
    for (hipEvent_t* ev : group_events(c)) if (*ev) { (void)hipEventDestroy(*ev); *ev = nullptr; }
    c.busy_valid = false;
    c.ready = false;
    c.device = -1;
    c.hip_dev = -1;
    return SBV_OK;
}
}  // namespace

extern "C" int sbv_shutdown(void) {
    std::unique_lock<std::shared_mutex> rl(g_reg_mu);
    g_reg = Registry();
    std::lock_guard<std::mutex> lk(g_mu);
    rccl_teardown();
    for (auto& up : g_ctxs) {
        if (!up) continue;
        std::lock_guard<std::mutex> lkc(up->mu);
        (void)shutdown_context(*up);
    }
    g_def = nullptr;
    return SBV_OK;
}

extern "C" int sbv_p256_verify_batch_dev(const void* d_tuples, size_t n, void* d_bitmap, void* hip_stream) {
    SBV_ENTER(c);
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!d_tuples || !d_bitmap || (reinterpret_cast<uintptr_t>(d_tuples) & 15)) {
        g_err = "null or misaligned device pointer";
        return SBV_EINVAL;
    }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    int rc = ensure_capacity(c, n < kMaxChunk ? n : kMaxChunk);
    if (rc != SBV_OK) return rc;
    // the scratch is single-flight: order this call after the previous one even across streams
    if (c.busy_valid) HIP_TRY(SBV_EDEVICE, hipStreamWaitEvent(stream, c.busy, 0));
    const uint8_t* src = static_cast<const uint8_t*>(d_tuples);
    uint8_t* dst = static_cast<uint8_t*>(d_bitmap);
    for (size_t off = 0; off < n; off += kMaxChunk) {       // kMaxChunk is a multiple of 8
        const size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        hipEvent_t mid = nullptr, end = nullptr;
        if (c.profiling == 1) {
            if (c.prof_used + 3 > c.prof_events.size()) {
                for (int k = 0; k < 3; ++k) {
                    hipEvent_t ev;
                    HIP_TRY(SBV_EDEVICE, hipEventCreate(&ev));
                    c.prof_events.push_back(ev);
                }
            }
            HIP_TRY(SBV_EDEVICE, hipEventRecord(c.prof_events[c.prof_used], stream));
            mid = c.prof_events[c.prof_used + 1];
            end = c.prof_events[c.prof_used + 2];
            c.prof_used += 3;
        }
        // event pairs around every launch of the dominant kernel (one per chunk of windows when grouped)
        hipEvent_t* dom = nullptr;
        int dom_pairs = 0;
        if (c.profiling) {
            while (c.prof_dom_used + 2 * SBV_GROUP_MAX_CHUNKS > c.prof_dom.size()) {
                hipEvent_t ev;
                HIP_TRY(SBV_EDEVICE, hipEventCreate(&ev));
                c.prof_dom.push_back(ev);
            }
            dom = c.prof_dom.data() + c.prof_dom_used;
        }
        bool was_grouped = false;
        rc = enqueue(c, src + off * SBV_TUPLE_BYTES, m, dst + off / 8, stream, mid, dom, &dom_pairs, &was_grouped);
        if (rc != SBV_OK) {          // part of the step may be enqueued: later users of the scratch must still wait for it
            if (hipEventRecord(c.busy, stream) == hipSuccess) c.busy_valid = true;
            return rc;
        }
        (void)was_grouped;
        if (c.profiling) c.prof_dom_used += 2 * (size_t)dom_pairs;
        if (end) HIP_TRY(SBV_EDEVICE, hipEventRecord(end, stream));
    }
    HIP_TRY(SBV_EDEVICE, hipEventRecord(c.busy, stream));
    c.busy_valid = true;
    return SBV_OK;
}

// The host-pointer entry a cgo caller uses, pipelined: two staging slots (device tuples + bitmap + page-locked bitmap) and a
// copy stream.  A call enqueues H2D (copy stream) -> stage A + B (c.stream, after the copy's event) -> D2H, records the
// slot's completion event and RELEASES the context lock before it waits — so the next caller (another goroutine's cgo
// thread, or this call's next chunk) uploads its tuples while these kernels run.  Kernels of all calls are ordered on
// c.stream, so the single-flight scratch and group buffers need no further care.
namespace {
struct Outstanding { int slot; size_t off, m; };

int collect_slot(Context& c, const Outstanding& o, uint8_t* accept_bitmap, sbv_timing& tm) {
    StageSlot& sl = c.stage[o.slot];
    const hipError_t e = hipEventSynchronize(sl.ev[4]);
    int rc = SBV_OK;
    if (e == hipSuccess) {
        memcpy(accept_bitmap + o.off / 8, sl.h_bitmap, (o.m + 7) / 8);
        tm.h2d_us += 1e3 * ms_between(sl.ev[0], sl.ev[1]);
        tm.prep_us += 1e3 * ms_between(sl.ev[1], sl.ev[2]);
        tm.verify_us += 1e3 * ms_between(sl.ev[2], sl.ev[3]);
        tm.d2h_us += 1e3 * ms_between(sl.ev[3], sl.ev[4]);
    } else {
        rc = fail(SBV_EDEVICE, "hipEventSynchronize(slot)", e);
    }
    {
        std::lock_guard<std::mutex> lk(c.mu);
        sl.used = false;
    }
    c.slot_cv.notify_all();
    return rc;
}

int grow_slot(Context& c, StageSlot& sl, size_t m) {
    if (!sl.ev[0]) for (auto& ev : sl.ev) HIP_TRY(SBV_EDEVICE, hipEventCreate(&ev));
    const size_t want = (m + 1023) & ~(size_t)1023;
    if (want <= sl.cap) return SBV_OK;
    if (sl.d_tuples) (void)hipFree(sl.d_tuples);
    if (sl.d_bitmap) (void)hipFree(sl.d_bitmap);
    if (sl.h_bitmap) (void)hipHostFree(sl.h_bitmap);
    sl.d_tuples = sl.d_bitmap = sl.h_bitmap = nullptr; sl.cap = 0;
    HIP_TRY(SBV_ENOMEM, hipMalloc(&sl.d_tuples, want * SBV_TUPLE_BYTES));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&sl.d_bitmap, want / 8));
    HIP_TRY(SBV_ENOMEM, hipHostMalloc(&sl.h_bitmap, want / 8, hipHostMallocDefault));
    sl.cap = want;
    (void)c;
    return SBV_OK;
}
}  // namespace

extern "C" int sbv_p256_verify_batch(const uint8_t* tuples, size_t n, uint8_t* accept_bitmap) {
    Context* cp = default_ctx();
    Context& c = *cp;
    if (n == 0) { std::lock_guard<std::mutex> lk(c.mu); if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; } return SBV_OK; }
    if (!tuples || !accept_bitmap) { g_err = "null pointer"; return SBV_EINVAL; }
    const auto t0 = std::chrono::steady_clock::now();
    // ONE caller with a large batch (VerifyProposal is one caller: internal/bft/view.go:555) used to upload, THEN verify: 152 M/s with
    // one submitting thread against 280 with two.  With the key-table cache on (the default) its batch is now cut into pieces of 2^18
    // tuples that take the two staging slots in turn — the upload of piece i + 1 runs beside the kernels of piece i, the first piece
    // builds the signers' combs, the later ones find them — through the same slot protocol concurrent callers share, so two callers
    // still interleave.  With the cache off every piece would rebuild every table: whole launches, as before.
    size_t piece = kMaxChunk;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        if (c.ready && n >= ((size_t)1 << 19) && c.kc_on[0] && c.group_enabled) piece = (size_t)1 << 18;
    }
    sbv_timing tm{};
    tm.n = n;
    std::vector<Outstanding> out;           // at most two
    int rc = SBV_OK;
    for (size_t off = 0; off < n && rc == SBV_OK; off += piece) {          // piece is a multiple of 8: whole bitmap bytes
        const size_t m = n - off < piece ? n - off : piece;
        std::unique_lock<std::mutex> lk(c.mu);
        if (!c.ready) { g_err = "sbv_init has not succeeded"; rc = SBV_ENOTINIT; break; }
        // a slot: never wait for one while holding one (two multi-chunk callers would deadlock) — collect first
        while (c.stage[0].used && c.stage[1].used) {
            if (!out.empty()) {
                const Outstanding o = out.front();
                out.erase(out.begin());
                lk.unlock();
                rc = collect_slot(c, o, accept_bitmap, tm);
                lk.lock();
                if (rc != SBV_OK) break;
            } else {
                c.slot_cv.wait(lk);
            }
        }
        if (rc != SBV_OK) break;
        if (hipSetDevice(c.hip_dev) != hipSuccess) { g_err = "hipSetDevice failed"; rc = SBV_EDEVICE; break; }
        if (((m + 1023) & ~(size_t)1023) > c.cap) {           // growing the scratch frees buffers in-flight work may use: drain first
            while (c.stage[0].used || c.stage[1].used) {
                if (!out.empty()) {
                    const Outstanding o = out.front();
                    out.erase(out.begin());
                    lk.unlock();
                    rc = collect_slot(c, o, accept_bitmap, tm);
                    lk.lock();
                    if (rc != SBV_OK) break;
                } else {
                    c.slot_cv.wait(lk);
                }
            }
            if (rc != SBV_OK) break;
            if ((rc = ensure_capacity(c, m)) != SBV_OK) break;
        }
        // The copy stream is created on first use: a process gets four hardware queues, and a fifth stream created at
        // sbv_init aliased one of the grouped step's side streams — the table kernels then queued behind the G phase
        // (measured: 3.9 -> 4.6 ms per 2^20 batch through the device-pointer entry, which never copies).
        if (!c.copy_stream && hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking) != hipSuccess) { g_err = "hipStreamCreate failed"; rc = SBV_EDEVICE; break; }
        const int s = c.stage[0].used ? 1 : 0;
        StageSlot& sl = c.stage[s];
        if ((rc = grow_slot(c, sl, m)) != SBV_OK) break;
        sl.used = true;
        hipError_t e = hipSuccess;
        auto step = [&](hipError_t r) { if (e == hipSuccess) e = r; };
        step(hipEventRecord(sl.ev[0], c.copy_stream));
        step(hipMemcpyAsync(sl.d_tuples, tuples + off * SBV_TUPLE_BYTES, m * SBV_TUPLE_BYTES, hipMemcpyHostToDevice, c.copy_stream));
        step(hipEventRecord(sl.ev[1], c.copy_stream));
        step(hipStreamWaitEvent(c.stream, sl.ev[1], 0));
        if (c.busy_valid) step(hipStreamWaitEvent(c.stream, c.busy, 0));
        if (e == hipSuccess) rc = enqueue(c, sl.d_tuples, m, sl.d_bitmap, c.stream, sl.ev[2]);
        step(hipEventRecord(sl.ev[3], c.stream));
        step(hipMemcpyAsync(sl.h_bitmap, sl.d_bitmap, (m + 7) / 8, hipMemcpyDeviceToHost, c.stream));
        step(hipEventRecord(sl.ev[4], c.stream));
        if (hipEventRecord(c.busy, c.stream) == hipSuccess) c.busy_valid = true;
        if (e != hipSuccess && rc == SBV_OK) rc = fail(SBV_EDEVICE, "enqueue of a staged chunk", e);
        if (rc != SBV_OK) {                  // whatever was enqueued must drain before the slot is reused
            (void)hipStreamSynchronize(c.copy_stream);
            (void)hipStreamSynchronize(c.stream);
            sl.used = false;
            lk.unlock();
            c.slot_cv.notify_all();
            break;
        }
        lk.unlock();
        out.push_back({s, off, m});
        if (out.size() == 2) {
            const Outstanding o = out.front();
            out.erase(out.begin());
            rc = collect_slot(c, o, accept_bitmap, tm);
        }
    }
    for (const Outstanding& o : out) {       // also on error: the slots must come back
        const int r2 = collect_slot(c, o, accept_bitmap, tm);
        if (rc == SBV_OK) rc = r2;
    }
    if (rc != SBV_OK) return rc;
    tm.total_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    std::lock_guard<std::mutex> lk(c.mu);
    c.timing = tm;
    return SBV_OK;
}

namespace {
// Wide combs for `slots` (those that have none yet), kwide_max at most in total: built on the host like the 8-bit ones (one thread
// per window, the same field code as the kernels), uploaded key by key.  16 bits: 35.7 MB and ~0.1 s per key; 20 bits: 436 MB
// and about a second.
constexpr size_t kWideAutoSplit = 16;
int drop_wide_keys(Context& c);
// A wide-comb pool may take HBM only while a reserve stays free for everything that is sized per batch (the scratch of a 2^21-tuple
// launch, the grouping arrays and comb pools of the three schemes: ~8 GB): min(16 GB, a quarter of the device).  On an MI355X
// (288 GB) the 7 GB of a 16-node cluster's 20-bit combs are far inside; on a smaller or crowded device the policy steps down to
// 16 bits and then to "no wide combs" instead of leaving later batches with SBV_ENOMEM.
// With several contexts on one GPU (SBV_LOGICAL_DEVICES) a pool may also take no more than 15 % of the device divided by the contexts
// that share it: 43 GB alone on a 288 GB part (the 1024-key hot pool is 36.5), 5.4 GB as one of eight.
bool wide_pool_fits(const Context& c, size_t extra_bytes, unsigned percent) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return true;
    const size_t reserve = total_b / 4 < ((size_t)16 << 30) ? total_b / 4 : ((size_t)16 << 30);
    size_t cap = total_b / 100 * percent / (size_t)(c.mem_share > 0 ? c.mem_share : 1);
    if (const char* e = getenv("SBV_POOL_BUDGET_MB")) { const long v = atol(e); if (v > 0) cap = (size_t)v << 20; }
    return extra_bytes <= cap && free_b > extra_bytes && free_b - extra_bytes >= reserve;
}
int widen_slots(Context& c, const std::vector<u32>& slots) {
    std::vector<u32> todo;
    {
        std::vector<uint8_t> has(c.nkeys, 0);
        for (u32 sl : c.wide_slots) if (sl < has.size()) has[sl] = 1;
        for (u32 sl : slots) {
            if (sl >= c.nkeys) { g_err = "sbv_p256_widen_keys: slot is not registered"; return SBV_EINVAL; }
            if (!has[sl] && c.wide_slots.size() + todo.size() < (size_t)c.kwide_max) { has[sl] = 1; todo.push_back(sl); }
        }
    }
    if (todo.empty()) return SBV_OK;
    c.wide_nofit = false;
    if (c.kwide_auto) {
        // The width follows the size of the consenter set: 20-bit combs (13 additions, 436 MB per key) while at most kWideAutoSplit
        // keys are wide — a 16-node cluster holds 7 GB of them — and 16-bit combs (16 additions, 35.7 MB) beyond; crossing the line
        // rebuilds what was there (two launches: milliseconds).
        int want_bits = c.wide_slots.size() + todo.size() <= kWideAutoSplit ? 20 : 16;
        // the 20-bit pool is sized ONCE for a full 16-key set (no doubling, no old + new copies side by side): does it fit?
        if (want_bits == 20 && !(c.kwide_bits == 20 && c.kwide_cap >= kWideAutoSplit) &&
            !wide_pool_fits(c, (kWideAutoSplit < c.kwide_max ? kWideAutoSplit : (size_t)c.kwide_max) * sbv::gcomb_entries(20) * sizeof(sbv::apt))) want_bits = 16;
        if (want_bits != c.kwide_bits) {
            // the pool of the NEW width must fit before anything is dropped (a memory-tight device used to lose the combs it had
            // and then find that the narrower pool did not fit either)
            const size_t ncap = want_bits >= 18 ? kWideAutoSplit : 64;
            const size_t need = (ncap < c.kwide_max ? ncap : (size_t)c.kwide_max) * sbv::gcomb_entries(want_bits) * sizeof(sbv::apt);
            const size_t held = c.kwide_cap * sbv::gcomb_entries(c.kwide_bits) * sizeof(sbv::apt);     // freed by the drop
            if (!wide_pool_fits(c, need > held ? need - held : 0)) { c.wide_nofit = true; return SBV_OK; }
            std::vector<u32> all = c.wide_slots;
            all.insert(all.end(), todo.begin(), todo.end());
            const int r = drop_wide_keys(c);
            if (r != SBV_OK) return r;
            c.kwide_bits = want_bits;
            todo = all;
        }
    }
    const size_t stride = sbv::gcomb_entries(c.kwide_bits);
    const size_t want = c.wide_slots.size() + todo.size();
    if (want > c.kwide_cap) {
        // 18-20-bit combs: a 16-key pool at once (the auto policy never holds more at 20 bits).  Narrower combs: 64 keys at once (2.3 GB at 16 bits).  Doubling beyond.
        size_t cap = c.kwide_cap ? c.kwide_cap : (c.kwide_bits >= 18 ? kWideAutoSplit : 64);
        while (cap < want) cap *= 2;
        if (cap > c.kwide_max) cap = c.kwide_max;
        if (!wide_pool_fits(c, cap * stride * sizeof(sbv::apt))) { c.wide_nofit = true; return SBV_OK; }      // no room beside the reserve: the slots keep their 8-bit combs (same verdicts)
        sbv::apt* nt = nullptr;
        HIP_TRY(SBV_ENOMEM, hipMalloc(&nt, cap * stride * sizeof(sbv::apt)));
        HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());          // nothing in flight may still read the old tables
        if (!c.wide_slots.empty()) {
            const hipError_t e = hipMemcpy(nt, c.d_kwide, c.wide_slots.size() * stride * sizeof(sbv::apt), hipMemcpyDeviceToDevice);
            if (e != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) { (void)hipFree(nt); return fail(SBV_EDEVICE, "wide combs: copy", e); }
        }
        if (c.d_kwide) (void)hipFree(c.d_kwide);
        c.d_kwide = nt;
        c.kwide_cap = cap;
    }
    std::vector<const std::string*> key_of(c.nkeys, nullptr);
    for (const auto& kv : c.key_index) if (kv.second < key_of.size()) key_of[kv.second] = &kv.first;
    // Built on the device (p256_widetab29.h): the host computes 2 * windows base points per key (microseconds), two launches fill
    // the combs of all the keys of this call.  SBV_KEYED_WIDE_HOST=1 keeps the host builder (one thread per window: 0.1 - 1.3 s per key).
    static const bool host_build = [] { const char* e = getenv("SBV_KEYED_WIDE_HOST"); return e && e[0] == '1'; }();
    const sbv::widebuild wb = sbv::widebuild_make(c.kwide_bits);
    std::vector<sbv::apt> bases;
    std::vector<u32> widx_of, slot_of_build;
    std::vector<sbv::apt> tab(host_build ? stride : 0);
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    const size_t first_new = c.wide_slots.size();
    // a failure below leaves the registry as it was before this call: the slots named here stay on their 8-bit combs and may be named again
    auto rollback = [&] {
        const u32 none = SBV_WIDE_NONE;
        for (size_t i = first_new; i < c.wide_slots.size(); ++i) (void)hipMemcpy(c.d_kwidx + c.wide_slots[i], &none, sizeof(u32), hipMemcpyHostToDevice);
        c.wide_slots.resize(first_new);
    };
    for (u32 sl : todo) {
        const std::string* k = key_of[sl];
        const u32 w = (u32)c.wide_slots.size();
        bool have = false;
        if (k && k->size() == 64) {
            if (host_build) {
                have = sbv::host_build_wide_key_table((const uint8_t*)k->data(), c.kwide_bits, tab.data(), (int)(hw > 32 ? 32 : hw));
                if (have) { const hipError_t e = hipMemcpy(c.d_kwide + (size_t)w * stride, tab.data(), stride * sizeof(sbv::apt), hipMemcpyHostToDevice); if (e != hipSuccess) { rollback(); return fail(SBV_EDEVICE, "wide combs: upload", e); } }
            } else {
                bases.resize(bases.size() + 2 * (size_t)wb.windows);
                have = sbv::host_wide_bases((const uint8_t*)k->data(), c.kwide_bits, bases.data() + bases.size() - 2 * (size_t)wb.windows);
                if (have) { widx_of.push_back(w); slot_of_build.push_back(sl); }
                else bases.resize(bases.size() - 2 * (size_t)wb.windows);
            }
        }
        // not a point: kvalid[slot] = 0 rejects its signatures whatever the lanes add; its comb is zeros
        if (!have) { const hipError_t e = memset_now(c.d_kwide + (size_t)w * stride, 0, stride * sizeof(sbv::apt)); if (e != hipSuccess) { rollback(); return fail(SBV_EDEVICE, "wide combs: zero", e); } }
        c.wide_slots.push_back(sl);
        if (host_build || !have) {      // published after its table is complete
            const hipError_t e = hipMemcpy(c.d_kwidx + sl, &w, sizeof(u32), hipMemcpyHostToDevice);
            if (e != hipSuccess) { rollback(); return fail(SBV_EDEVICE, "wide combs: publish", e); }
        }
    }
    if (!widx_of.empty()) {
        const u32 nb = (u32)widx_of.size();
        sbv::apt* d_bases = nullptr;
        u32* d_w = nullptr;
        u32* d_tmp = nullptr;
        int rc = SBV_OK;
        auto cleanup = [&] { if (d_bases) (void)hipFree(d_bases); if (d_w) (void)hipFree(d_w); if (d_tmp) (void)hipFree(d_tmp); };
        hipError_t e = hipMalloc(&d_bases, bases.size() * sizeof(sbv::apt));
        if (e == hipSuccess) e = hipMalloc(&d_w, nb * sizeof(u32));
        if (e == hipSuccess) e = hipMalloc(&d_tmp, sbv::widetab_tmp_words(nb, c.kwide_bits) * sizeof(u32));
        if (e != hipSuccess) rc = fail(SBV_ENOMEM, "wide combs: scratch", e);
        if (rc == SBV_OK) {
            e = hipMemcpy(d_bases, bases.data(), bases.size() * sizeof(sbv::apt), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(d_w, widx_of.data(), nb * sizeof(u32), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = sbv::launch_widetab_build(d_bases, d_w, nb, c.kwide_bits, d_tmp, c.d_kwide, c.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
            if (e != hipSuccess) rc = fail(SBV_EDEVICE, "wide combs: build", e);
        }
        if (rc == SBV_OK)        // published after the tables are complete
            for (u32 i = 0; i < nb && rc == SBV_OK; ++i) {
                e = hipMemcpy(c.d_kwidx + slot_of_build[i], &widx_of[i], sizeof(u32), hipMemcpyHostToDevice);
                if (e != hipSuccess) rc = fail(SBV_EDEVICE, "wide combs: publish", e);
            }
        cleanup();
        if (rc != SBV_OK) { rollback(); return rc; }
    }
    return SBV_OK;
}
int drop_wide_keys(Context& c) {
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    if (c.d_kwidx && c.key_cap) HIP_TRY(SBV_EDEVICE, memset_now(c.d_kwidx, 0xFF, c.key_cap * sizeof(u32)));
    if (c.d_kwide) (void)hipFree(c.d_kwide);
    c.d_kwide = nullptr; c.kwide_cap = 0; c.wide_slots.clear();
    return SBV_OK;
}
}  // namespace

extern "C" int sbv_p256_wide_keys(int bits, uint32_t max_keys) {
    if (bits != 0 && bits != SBV_WIDE_BITS_AUTO && (bits < 10 || bits > 20)) return SBV_EINVAL;
    if (max_keys > 4096) return SBV_EINVAL;
    std::unique_lock<std::shared_mutex> rl(g_reg_mu);
    {
        std::lock_guard<std::mutex> lk(g_set_mu);
        if (bits) g_settings.wide_bits = bits;
        g_settings.wide_max = bits ? max_keys : 0u;
    }
    if (g_reg.wide.size() > (bits ? max_keys : 0u)) g_reg.wide.resize(bits ? max_keys : 0u);      // what every device keeps (the first ones named)
    int rc = SBV_OK;
    for (Context* cp : live_contexts()) {
        std::lock_guard<std::mutex> lk(cp->mu);
        Context& c = *cp;
        if (!c.ready) continue;
        if (hipSetDevice(c.hip_dev) != hipSuccess) { rc = SBV_EDEVICE; continue; }
        const u32 nmax = bits ? max_keys : 0u;
        std::vector<u32> had = c.wide_slots;
        if (had.size() > nmax) had.resize(nmax);
        int want_bits = c.kwide_bits;
        if (bits == SBV_WIDE_BITS_AUTO) want_bits = had.size() <= kWideAutoSplit ? 20 : 16;
        else if (bits) want_bits = bits;
        const bool rebuild = want_bits != c.kwide_bits || nmax < c.wide_slots.size();       // another width, or fewer keys than it holds
        if (rebuild) { const int r = drop_wide_keys(c); if (r != SBV_OK) { rc = r; continue; } }
        if (bits) { c.kwide_auto = bits == SBV_WIDE_BITS_AUTO; c.kwide_bits = want_bits; }
        c.kwide_max = nmax;
        if (rebuild && nmax) { const int r = widen_slots(c, had); if (r != SBV_OK) rc = r; }
    }
    return rc;
}

namespace {
// c.mu held, the device current.  Appends `count` keys (slots c.nkeys, c.nkeys + 1, ...) with their host-built 8-bit combs.  The
// index is only extended AFTER both uploads succeeded: an entry left behind by a failed call would hand its slot number to the
// next fresh key, and the first key's ID would then resolve to another signer's comb.
int append_keys(Context& c, const std::string* keys, const sbv::apt* tabs, const uint8_t* valid, size_t count) {
    if (count == 0) return SBV_OK;
    const int rc = ensure_key_capacity(c, c.nkeys + count);
    if (rc != SBV_OK) return rc;
    HIP_TRY(SBV_EDEVICE, hipMemcpy(c.d_ktab + c.nkeys * (size_t)SBV_KEYTAB_ENTRIES, tabs, count * (size_t)SBV_KEYTAB_ENTRIES * sizeof(sbv::apt), hipMemcpyHostToDevice));
    HIP_TRY(SBV_EDEVICE, hipMemcpy(c.d_kvalid + c.nkeys, valid, count, hipMemcpyHostToDevice));
    for (size_t i = 0; i < count; ++i) c.key_index.emplace(keys[i], (u32)(c.nkeys + i));
    c.nkeys += count;
    return SBV_OK;
}
// the 8-bit combs of `count` keys, built on the host (one-time setup, the same field code as the kernels), in parallel
void build_key_tables(const std::string* keys, size_t count, std::vector<sbv::apt>& tabs, std::vector<uint8_t>& valid) {
    tabs.resize(count * (size_t)SBV_KEYTAB_ENTRIES);
    valid.assign(count, 0);
    size_t nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 64) nt = 64;
    if (nt > count) nt = count;
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            for (size_t j = t; j < count; j += nt)
                valid[j] = sbv::host_build_key_table((const uint8_t*)keys[j].data(), &tabs[j * (size_t)SBV_KEYTAB_ENTRIES]) ? 1 : 0;
        });
    for (auto& x : th) x.join();
}
// g_reg_mu (shared or exclusive) and c.mu held.  Brings device c's replica of the registry up to the process-wide one: the 8-bit
// combs of the slots it has not seen, then the wide combs of the widened slots it lacks (built on the device: milliseconds).
int sync_registry(Context& c) {
    if (!c.ready) { g_err = "device not initialised"; return SBV_ENOTINIT; }
    const bool wide_done = c.wide_slots.size() >= (g_reg.wide.size() < (size_t)c.kwide_max ? g_reg.wide.size() : (size_t)c.kwide_max) ||
                           (c.wide_nofit && c.wide_nofit_at == g_reg.wide.size());      // "does not fit" stands until the list of widened slots changes
    if (c.nkeys == g_reg.keys.size() && wide_done) return SBV_OK;
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    if (c.nkeys > g_reg.keys.size()) { g_err = "device registry ahead of the process registry"; return SBV_EDEVICE; }
    if (c.nkeys < g_reg.keys.size()) {
        const size_t first = c.nkeys, count = g_reg.keys.size() - first;
        std::vector<sbv::apt> tabs;
        std::vector<uint8_t> valid;
        build_key_tables(g_reg.keys.data() + first, count, tabs, valid);
        const int rc = append_keys(c, g_reg.keys.data() + first, tabs.data(), valid.data(), count);
        if (rc != SBV_OK) return rc;
    }
    if (g_reg.wide.empty() || wide_done) return SBV_OK;
    const int wrc = widen_slots(c, g_reg.wide);      // skips the slots that are wide already, keeps the order
    if (c.wide_nofit) c.wide_nofit_at = g_reg.wide.size();
    return wrc;
}
}  // namespace

extern "C" int sbv_p256_widen_keys(const uint32_t* slots, size_t m) {
    std::unique_lock<std::shared_mutex> rl(g_reg_mu);
    Context* def = default_ctx();
    {
        std::lock_guard<std::mutex> lk(def->mu);
        Context& c = *def;
        if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
        if (m == 0) return SBV_OK;
        if (!slots) { g_err = "null pointer"; return SBV_EINVAL; }
        HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
        const int rc = widen_slots(c, std::vector<u32>(slots, slots + m));
        if (rc != SBV_OK) return rc;
    }
    for (size_t i = 0; i < m; ++i)
        if (std::find(g_reg.wide.begin(), g_reg.wide.end(), slots[i]) == g_reg.wide.end()) g_reg.wide.push_back(slots[i]);
    // the other devices of the node: best effort now, retried by the sharded registered-key entry
    for (Context* cp : live_contexts()) {
        if (cp == def) continue;
        std::lock_guard<std::mutex> lk(cp->mu);
        if (cp->ready) (void)sync_registry(*cp);
    }
    return SBV_OK;
}

// Diagnostics: is the device-resident wide comb of `slot` byte for byte what the host builder (the kernels' field code on the CPU,
// one thread per window) produces for that key?  1 = equal, 0 = different, < 0 = error (not widened, device fault).
extern "C" int sbv_p256_wide_selfcheck(uint32_t slot) {
    SBV_ENTER(c);
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    size_t w = c.wide_slots.size();
    for (size_t i = 0; i < c.wide_slots.size(); ++i) if (c.wide_slots[i] == slot) w = i;
    if (w == c.wide_slots.size()) { g_err = "sbv_p256_wide_selfcheck: the slot has no wide comb"; return SBV_EINVAL; }
    const std::string* key = nullptr;
    for (const auto& kv : c.key_index) if (kv.second == slot) key = &kv.first;
    if (!key) return SBV_EINVAL;
    const size_t stride = sbv::gcomb_entries(c.kwide_bits);
    std::vector<sbv::apt> want(stride), got(stride);
    unsigned hw = std::thread::hardware_concurrency();
    if (!sbv::host_build_wide_key_table((const uint8_t*)key->data(), c.kwide_bits, want.data(), (int)(hw > 32 ? 32 : (hw ? hw : 1))))
        memset((void*)want.data(), 0, stride * sizeof(sbv::apt));
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    HIP_TRY(SBV_EDEVICE, hipMemcpy(got.data(), c.d_kwide + w * stride, stride * sizeof(sbv::apt), hipMemcpyDeviceToHost));
    return memcmp(got.data(), want.data(), stride * sizeof(sbv::apt)) == 0 ? 1 : 0;
}

extern "C" int sbv_p256_wide_key_stats(uint32_t out[4]) {
    SBV_ENTER(c);
    if (!c.ready) return SBV_ENOTINIT;
    if (!out) return SBV_EINVAL;
    out[0] = (u32)c.wide_slots.size(); out[1] = (u32)c.kwide_bits; out[2] = c.kwide_max;
    out[3] = (u32)((sbv::gcomb_entries(c.kwide_bits) * sizeof(sbv::apt)) >> 10);      // KiB per key
    return SBV_OK;
}

extern "C" int sbv_p256_register_keys(const uint8_t* keys, size_t m, uint32_t* slots_out) {
    std::unique_lock<std::shared_mutex> rl(g_reg_mu);
    Context* def = default_ctx();
    // the other devices' contexts, looked up BEFORE the default context is locked: the documented order is g_reg_mu -> g_mu -> Context::mu
    // (live_contexts() takes g_mu, and used to be called with def->mu held)
    const std::vector<Context*> contexts = live_contexts();
    std::vector<std::string> fresh;                 // keys that need a slot, in slot order
    {
        std::lock_guard<std::mutex> lk(def->mu);
        Context& c = *def;
        if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
        if (m == 0) return SBV_OK;
        if (!keys || !slots_out) { g_err = "null pointer"; return SBV_EINVAL; }
        HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
        if (c.nkeys != g_reg.keys.size()) {           // an earlier replication of the default device failed half-way: finish it first
            const int rc = sync_registry(c);
            if (rc != SBV_OK) return rc;
        }
        // de-duplicate, assign slots from the process-wide index
        std::unordered_map<std::string, u32> pending;
        for (size_t i = 0; i < m; ++i) {
            const std::string k((const char*)keys + 64 * i, 64);
            auto it = g_reg.index.find(k);
            if (it != g_reg.index.end()) { slots_out[i] = it->second; continue; }
            auto pt = pending.find(k);
            if (pt != pending.end()) { slots_out[i] = pt->second; continue; }
            const u32 slot = (u32)(g_reg.keys.size() + fresh.size());
            pending.emplace(k, slot);
            fresh.push_back(k);
            slots_out[i] = slot;
        }
        if (fresh.empty()) return SBV_OK;
        std::vector<sbv::apt> tabs;
        std::vector<uint8_t> valid;
        build_key_tables(fresh.data(), fresh.size(), tabs, valid);
        // the default device first: its failure fails the call and leaves the registry as it was
        const int rc = append_keys(c, fresh.data(), tabs.data(), valid.data(), fresh.size());
        if (rc != SBV_OK) return rc;
        for (const std::string& k : fresh) { g_reg.index.emplace(k, (u32)g_reg.keys.size()); g_reg.keys.push_back(k); }
        // the other devices of the node take the same tables (built once): best effort now, retried by the sharded registered-key entry
        for (Context* cp : contexts) {
            if (cp == def) continue;
            std::lock_guard<std::mutex> lk2(cp->mu);
            if (!cp->ready || hipSetDevice(cp->hip_dev) != hipSuccess) continue;
            if (cp->nkeys + fresh.size() == g_reg.keys.size()) (void)append_keys(*cp, fresh.data(), tabs.data(), valid.data(), fresh.size());
            else (void)sync_registry(*cp);
        }
        (void)hipSetDevice(c.hip_dev);
    }
    return SBV_OK;
}

extern "C" int sbv_p256_key_count(void) {
    SBV_ENTER(c);
    return c.ready ? (int)c.nkeys : SBV_ENOTINIT;
}

extern "C" int sbv_p256_clear_keys(void) {
    std::unique_lock<std::shared_mutex> rl(g_reg_mu);
    Context* def = default_ctx();
    { std::lock_guard<std::mutex> lk(def->mu); if (!def->ready) return SBV_ENOTINIT; }
    int rc = SBV_OK;
    for (Context* cp : live_contexts()) {
        std::lock_guard<std::mutex> lk(cp->mu);
        Context& c = *cp;
        if (!c.ready) continue;
        hipError_t e = hipSetDevice(c.hip_dev);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        c.key_index.clear();
        c.nkeys = 0;
        if (e == hipSuccess && c.d_kwidx && c.key_cap) e = memset_now(c.d_kwidx, 0xFF, c.key_cap * sizeof(u32));
        c.wide_slots.clear();        // the allocation stays for the next registry
        if (e != hipSuccess) rc = fail(SBV_EDEVICE, "sbv_p256_clear_keys", e);
    }
    g_reg = Registry();
    return rc;
}

extern "C" int sbv_p256_verify_batch_keyed_dev(const void* d_rsh, const void* d_slots, size_t n, void* d_bitmap, void* hip_stream) {
    SBV_ENTER(c);
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!d_rsh || !d_slots || !d_bitmap || (reinterpret_cast<uintptr_t>(d_rsh) & 15)) { g_err = "null or misaligned device pointer"; return SBV_EINVAL; }
    if (c.nkeys == 0) { g_err = "no keys registered"; return SBV_EINVAL; }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    int rc = ensure_capacity(c, n < kMaxChunk ? n : kMaxChunk);
    if (rc != SBV_OK) return rc;
    if (c.busy_valid) HIP_TRY(SBV_EDEVICE, hipStreamWaitEvent(stream, c.busy, 0));
    const uint8_t* src = static_cast<const uint8_t*>(d_rsh);
    const u32* sl = static_cast<const u32*>(d_slots);
    uint8_t* dst = static_cast<uint8_t*>(d_bitmap);
    for (size_t off = 0; off < n; off += kMaxChunk) {
        const size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        hipEvent_t mid = nullptr, end = nullptr;
        if (c.profiling) {
            if (c.prof_used + 3 > c.prof_events.size())
                for (int k = 0; k < 3; ++k) { hipEvent_t ev; HIP_TRY(SBV_EDEVICE, hipEventCreate(&ev)); c.prof_events.push_back(ev); }
            HIP_TRY(SBV_EDEVICE, hipEventRecord(c.prof_events[c.prof_used], stream));
            mid = c.prof_events[c.prof_used + 1];
            end = c.prof_events[c.prof_used + 2];
            c.prof_used += 3;
        }
        rc = enqueue_keyed(c, src + off * 96, sl + off, m, dst + off / 8, stream, mid);
        if (rc != SBV_OK) {
            if (hipEventRecord(c.busy, stream) == hipSuccess) c.busy_valid = true;
            return rc;
        }
        if (end) HIP_TRY(SBV_EDEVICE, hipEventRecord(end, stream));
    }
    HIP_TRY(SBV_EDEVICE, hipEventRecord(c.busy, stream));
    c.busy_valid = true;
    return SBV_OK;
}

extern "C" int sbv_p256_verify_batch_keyed(const uint8_t* rsh, const uint32_t* slots, size_t n, uint8_t* accept_bitmap) {
    SBV_ENTER(c);
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!rsh || !slots || !accept_bitmap) { g_err = "null pointer"; return SBV_EINVAL; }
    if (c.nkeys == 0) { g_err = "no keys registered"; return SBV_EINVAL; }
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    if (n <= SBV_SMALL_MAX && c.small_enabled) {
        // The latency form: ONE launch, no staging copies.  Stage A runs here on the host (host_prep_small), its records go into
        // a page-locked buffer the kernel reads over PCIe, every verdict comes back as a byte in mapped host memory, and this
        // thread polls the completion counter instead of sleeping in a stream synchronisation (a commit quorum: 15
        // signatures, internal/bft/view.go:531-541).
        if (!c.h_small_in) {
            HIP_TRY(SBV_ENOMEM, hipHostMalloc(&c.h_small_in, SBV_SMALL_MAX * (96 + 4), hipHostMallocMapped));
            HIP_TRY(SBV_ENOMEM, hipHostMalloc(&c.h_small_out, SBV_SMALL_MAX + 64, hipHostMallocMapped));
            HIP_TRY(SBV_EDEVICE, hipHostGetDevicePointer(&c.d_small_in, c.h_small_in, 0));
            HIP_TRY(SBV_EDEVICE, hipHostGetDevicePointer(&c.d_small_out, c.h_small_out, 0));
        }
        // stage A of the call on this thread (one inversion for all n signatures), straight into the mapped buffer
        sbv::host_prep_small(rsh, slots, n, reinterpret_cast<u32*>(c.h_small_in), reinterpret_cast<u32*>(c.h_small_in + SBV_SMALL_MAX * 96));
        volatile u32* done = reinterpret_cast<volatile u32*>(c.h_small_out + SBV_SMALL_MAX);
        *done = 0;
        std::atomic_thread_fence(std::memory_order_seq_cst);
        HIP_TRY(SBV_EDEVICE, sbv::launch_p256_verify_prepared_small(c.d_small_in, n, (u32)c.nkeys, c.d_ktab, c.d_kvalid, sbv::gcomb_make(c.d_g16r, c.g_bits),
                                                                  wide_of(c), static_cast<uint8_t*>(c.d_small_out),
                                                                  reinterpret_cast<u32*>(static_cast<uint8_t*>(c.d_small_out) + SBV_SMALL_MAX), c.stream));
        const auto give_up = t0 + std::chrono::milliseconds(5);
        while (*done < (u32)n) {
            if (std::chrono::steady_clock::now() > give_up) break;     // slow box / fault: let the runtime tell which
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
        }
        if (*done < (u32)n) HIP_TRY(SBV_EDEVICE, hipStreamSynchronize(c.stream));
        else (void)hipStreamQuery(c.stream);       // never blocks; lets the runtime retire the finished launch (this path never synchronises)
        std::atomic_thread_fence(std::memory_order_seq_cst);
        if (*done < (u32)n) { g_err = "the small-batch kernel did not report completion"; return SBV_EDEVICE; }
        memset(accept_bitmap, 0, (n + 7) / 8);
        for (size_t i = 0; i < n; ++i) if (c.h_small_out[i]) accept_bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));
        sbv_timing tm{};
        tm.n = n;
        tm.total_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        tm.verify_us = tm.total_us;
        c.timing = tm;
        return SBV_OK;
    }
    int rc = ensure_capacity(c, n < kMaxChunk ? n : kMaxChunk);
    if (rc != SBV_OK) return rc;
    if (c.busy_valid) HIP_TRY(SBV_EDEVICE, hipStreamWaitEvent(c.stream, c.busy, 0));
    sbv_timing tm{};
    tm.n = n;
    for (size_t off = 0; off < n; off += kMaxChunk) {
        const size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[0], c.stream));
        HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.d_tuples, rsh + off * 96, m * 96, hipMemcpyHostToDevice, c.stream));
        HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.d_slots, slots + off, m * sizeof(u32), hipMemcpyHostToDevice, c.stream));
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[1], c.stream));
        rc = enqueue_keyed(c, c.d_tuples, c.d_slots, m, c.d_bitmap, c.stream, c.ev[2]);
        if (rc != SBV_OK) return rc;
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[3], c.stream));
        HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.h_bitmap, c.d_bitmap, (m + 7) / 8, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[4], c.stream));
        HIP_TRY(SBV_EDEVICE, hipStreamSynchronize(c.stream));
        memcpy(accept_bitmap + off / 8, c.h_bitmap, (m + 7) / 8);
        tm.h2d_us += 1e3 * ms_between(c.ev[0], c.ev[1]);
        tm.prep_us += 1e3 * ms_between(c.ev[1], c.ev[2]);
        tm.verify_us += 1e3 * ms_between(c.ev[2], c.ev[3]);
        tm.d2h_us += 1e3 * ms_between(c.ev[3], c.ev[4]);
    }
    c.busy_valid = false;
    tm.total_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    c.timing = tm;
    return SBV_OK;
}

namespace {
std::vector<sbv::aniels> g_h_ed_b16, g_h_ed_bcomb;     // built once per process, uploaded to each context on its first Ed25519 call
std::once_flag g_ed_b16_once, g_ed_bcomb_once;
int g_ed_bbits = 20;
int ensure_ed_table(Context& c) {
    if (c.d_btab) return SBV_OK;
    std::call_once(g_ed_b16_once, [] {
        g_h_ed_b16.resize(SBV_ED_B16_ENTRIES);          // 16-bit comb of B: 50 MB, built by 16 host threads in ~0.2 s
        sbv::host_build_ed_b16(g_h_ed_b16.data());
    });
    HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_btab, g_h_ed_b16.size() * sizeof(sbv::aniels)));
    HIP_TRY(SBV_EDEVICE, hipMemcpy(c.d_btab, g_h_ed_b16.data(), g_h_ed_b16.size() * sizeof(sbv::aniels), hipMemcpyHostToDevice));
    return SBV_OK;
}
// the grouped step's comb of B (ed25519_group.h: edcomb)
int ensure_ed_bcomb(Context& c) {
    if (c.d_ed_bcomb) return SBV_OK;
    std::call_once(g_ed_bcomb_once, [] {
        if (const char* e = getenv("SBV_ED_B_BITS")) { const int v = atoi(e); if (v >= 12 && v <= 22) g_ed_bbits = v; }
        if (g_ed_bbits == 16) return;                   // the one-lane kernel's table serves
        g_h_ed_bcomb.resize(sbv::edcomb_entries(g_ed_bbits));
        sbv::host_build_ed_bcomb(g_ed_bbits, g_h_ed_bcomb.data());
    });
    c.ed_bbits = g_ed_bbits;
    c.ed_bpitch = (u32)sizeof(sbv::aniels);
    if (g_ed_bbits == 16) { c.d_ed_bcomb = c.d_btab; return SBV_OK; }
    // one 128-byte line per entry (ed25519_core.h: edcomb::pitch; SBV_ED_B_PITCH=96 keeps the packed layout for A/B runs): 872 MB instead of 654
    u32 pitch = 128;
    if (const char* e = getenv("SBV_ED_B_PITCH")) { if (atoi(e) == 96) pitch = 96; }
    HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_ed_bcomb, g_h_ed_bcomb.size() * (size_t)pitch));
    HIP_TRY(SBV_EDEVICE, hipMemcpy2D(c.d_ed_bcomb, pitch, g_h_ed_bcomb.data(), sizeof(sbv::aniels), sizeof(sbv::aniels), g_h_ed_bcomb.size(), hipMemcpyHostToDevice));
    c.ed_bpitch = pitch;
    return SBV_OK;
}
}  // namespace

extern "C" int sbv_ed25519_verify_batch_dev(const void* d_tuples, size_t n, void* d_bitmap, void* hip_stream) {
    SBV_ENTER(c);
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!d_tuples || !d_bitmap || (reinterpret_cast<uintptr_t>(d_tuples) & 15)) { g_err = "null or misaligned device pointer"; return SBV_EINVAL; }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    int rc = ensure_capacity(c, n < kMaxChunk ? n : kMaxChunk);
    if (rc != SBV_OK) return rc;
    if ((rc = ensure_ed_table(c)) != SBV_OK) return rc;
    if (c.busy_valid) HIP_TRY(SBV_EDEVICE, hipStreamWaitEvent(stream, c.busy, 0));
    const uint8_t* src = static_cast<const uint8_t*>(d_tuples);
    uint8_t* dst = static_cast<uint8_t*>(d_bitmap);
    for (size_t off = 0; off < n; off += kMaxChunk) {
        const size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        hipEvent_t end = nullptr;
        if (c.profiling) {
            if (c.prof_used + 3 > c.prof_events.size())
                for (int k = 0; k < 3; ++k) { hipEvent_t ev; HIP_TRY(SBV_EDEVICE, hipEventCreate(&ev)); c.prof_events.push_back(ev); }
            HIP_TRY(SBV_EDEVICE, hipEventRecord(c.prof_events[c.prof_used], stream));
            HIP_TRY(SBV_EDEVICE, hipEventRecord(c.prof_events[c.prof_used + 1], stream));     // no stage A
            end = c.prof_events[c.prof_used + 2];
            c.prof_used += 3;
        }
        hipEvent_t* dom = nullptr;                // event pairs around the dominant kernel (k_ed_qphase, one launch per chunk of windows)
        int dom_pairs = 0;
        if (c.profiling) {
            while (c.prof_dom_used + 2 * SBV_GROUP_MAX_CHUNKS > c.prof_dom.size()) { hipEvent_t ev; HIP_TRY(SBV_EDEVICE, hipEventCreate(&ev)); c.prof_dom.push_back(ev); }
            dom = c.prof_dom.data() + c.prof_dom_used;
        }
        if ((rc = enqueue_ed25519(c, src + off * 128, m, dst + off / 8, stream, dom, &dom_pairs)) != SBV_OK) {
            if (hipEventRecord(c.busy, stream) == hipSuccess) c.busy_valid = true;
            return rc;
        }
        if (c.profiling) c.prof_dom_used += 2 * (size_t)dom_pairs;
        if (end) HIP_TRY(SBV_EDEVICE, hipEventRecord(end, stream));
    }
    HIP_TRY(SBV_EDEVICE, hipEventRecord(c.busy, stream));
    c.busy_valid = true;
    return SBV_OK;
}

extern "C" int sbv_ed25519_verify_batch(const uint8_t* tuples, size_t n, uint8_t* accept_bitmap) {
    SBV_ENTER(c);
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!tuples || !accept_bitmap) { g_err = "null pointer"; return SBV_EINVAL; }
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    int rc = ensure_capacity(c, n < kMaxChunk ? n : kMaxChunk);
    if (rc != SBV_OK) return rc;
    if ((rc = ensure_ed_table(c)) != SBV_OK) return rc;
    if (c.busy_valid) HIP_TRY(SBV_EDEVICE, hipStreamWaitEvent(c.stream, c.busy, 0));
    sbv_timing tm{};
    tm.n = n;
    for (size_t off = 0; off < n; off += kMaxChunk) {
        const size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[0], c.stream));
        HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.d_tuples, tuples + off * 128, m * 128, hipMemcpyHostToDevice, c.stream));
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[1], c.stream));
        if ((rc = enqueue_ed25519(c, c.d_tuples, m, c.d_bitmap, c.stream)) != SBV_OK) return rc;
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[3], c.stream));
        HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.h_bitmap, c.d_bitmap, (m + 7) / 8, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[4], c.stream));
        HIP_TRY(SBV_EDEVICE, hipStreamSynchronize(c.stream));
        memcpy(accept_bitmap + off / 8, c.h_bitmap, (m + 7) / 8);
        tm.h2d_us += 1e3 * ms_between(c.ev[0], c.ev[1]);
        tm.verify_us += 1e3 * ms_between(c.ev[1], c.ev[3]);
        tm.d2h_us += 1e3 * ms_between(c.ev[3], c.ev[4]);
    }
    c.busy_valid = false;
    tm.total_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    c.timing = tm;
    return SBV_OK;
}

// ---- secp256k1 variant ("other curves") ------------------------------------------------------
namespace {
std::vector<sbv::kapt> g_h_k256_gtab;           // built once per process, uploaded to each context on its first secp256k1 call
std::once_flag g_k256_once;
std::vector<sbv::kapt> g_h_k256_gcomb;          // the grouped step's comb, built once per process
int g_k256_gbits = 20;
std::once_flag g_k256_gcomb_once;
int ensure_k256_gcomb(Context& c) {
    if (c.d_k256_gcomb) return SBV_OK;
    std::call_once(g_k256_gcomb_once, [] {
        if (const char* e = getenv("SBV_K256_G_BITS")) { const int v = atoi(e); if (v >= 12 && v <= 22) g_k256_gbits = v; }
        if (g_k256_gbits == 16) return;                 // the one-lane kernel's table serves
        g_h_k256_gcomb.resize((size_t)((257 + g_k256_gbits - 1) / g_k256_gbits) << (g_k256_gbits - 1));
        sbv::host_build_k256_gcomb(g_k256_gbits, g_h_k256_gcomb.data());
    });
    c.k256_gbits = g_k256_gbits;
    if (g_k256_gbits == 16) { c.d_k256_gcomb = c.d_k256_gtab; return SBV_OK; }
    HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_k256_gcomb, g_h_k256_gcomb.size() * sizeof(sbv::kapt)));
    HIP_TRY(SBV_EDEVICE, hipMemcpy(c.d_k256_gcomb, g_h_k256_gcomb.data(), g_h_k256_gcomb.size() * sizeof(sbv::kapt), hipMemcpyHostToDevice));
    return SBV_OK;
}
int ensure_k256_table(Context& c) {
    if (c.d_k256_gtab) return SBV_OK;
    std::call_once(g_k256_once, [] {
        g_h_k256_gtab.resize(SBV_K256_GTABLE_ENTRIES);
        sbv::host_build_k256_gtable(g_h_k256_gtab.data());     // 17 host threads, a fraction of a second
    });
    HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_k256_gtab, g_h_k256_gtab.size() * sizeof(sbv::kapt)));
    HIP_TRY(SBV_EDEVICE, hipMemcpy(c.d_k256_gtab, g_h_k256_gtab.data(), g_h_k256_gtab.size() * sizeof(sbv::kapt), hipMemcpyHostToDevice));
    return SBV_OK;
}
}  // namespace

namespace {
// one chunk (m <= cap) of secp256k1 tuples on `stream`: the grouped step (k256_group_kernels.hip) or the one-lane kernel
int enqueue_k256(Context& c, const uint8_t* d_tuples, size_t m, uint8_t* d_bitmap, hipStream_t stream, hipEvent_t* dom = nullptr, int* dom_pairs = nullptr) {
    const sbv::Scratch s = scratch_view(c);
    bool grouped = c.group_enabled && m >= (c.kc_on[1] ? c.group_min_batch : c.group_min_batch_k256);
    if (grouped && !group_buffers_or_fallback(c, m, [&] { const int r = ensure_k256_group_buffers(c, m); return r != SBV_OK ? r : ensure_k256_gcomb(c); }, grouped)) return g_last_rc;
    if (grouped) {
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.gsync.ev_fork, stream));
        sbv::GroupSync y = c.gsync;                 // second table stream: the context's own, when the caller's runs the step (enqueue() says why)
        if (y.tstreams > 1 && stream != c.stream) y.side_t = c.stream;
        else y.tstreams = 1;
        const hipError_t ge = sbv::launch_k256_verify_grouped(d_tuples, s, m, variant_view(c, m), c.k256pool, c.d_qtab, c.d_k256_gtab, c.d_k256_gcomb, c.k256_gbits, d_bitmap, stream, y, dom, dom_pairs);
        if (ge != hipSuccess) {
            (void)hipDeviceSynchronize();
            (void)key_cache_forget(c.k256pool.kc);
            return fail(SBV_EDEVICE, "launch_k256_verify_grouped", ge);
        }
        return SBV_OK;
    }
    HIP_TRY(SBV_EDEVICE, sbv::launch_k256_verify(d_tuples, m, s, c.d_qtab, c.d_k256_gtab, d_bitmap, stream));
    return SBV_OK;
}
}  // namespace

extern "C" int sbv_secp256k1_verify_batch_dev(const void* d_tuples, size_t n, void* d_bitmap, void* hip_stream) {
    SBV_ENTER(c);
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!d_tuples || !d_bitmap || (reinterpret_cast<uintptr_t>(d_tuples) & 15)) { g_err = "null or misaligned device pointer"; return SBV_EINVAL; }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    int rc = ensure_capacity(c, n < kMaxChunk ? n : kMaxChunk);
    if (rc != SBV_OK) return rc;
    if ((rc = ensure_k256_table(c)) != SBV_OK) return rc;
    if (c.busy_valid) HIP_TRY(SBV_EDEVICE, hipStreamWaitEvent(stream, c.busy, 0));
    const uint8_t* src = static_cast<const uint8_t*>(d_tuples);
    uint8_t* dst = static_cast<uint8_t*>(d_bitmap);
    for (size_t off = 0; off < n && rc == SBV_OK; off += kMaxChunk) {
        const size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        hipEvent_t* dom = nullptr;                // event pairs around the dominant kernel (k_k256_qphase), read by sbv_profile_read_dominant
        int dom_pairs = 0;
        if (c.profiling) {
            while (c.prof_dom_used + 2 * SBV_GROUP_MAX_CHUNKS > c.prof_dom.size()) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) { g_err = "hipEventCreate failed"; return SBV_EDEVICE; } c.prof_dom.push_back(ev); }
            dom = c.prof_dom.data() + c.prof_dom_used;
        }
        rc = enqueue_k256(c, src + off * 160, m, dst + off / 8, stream, dom, &dom_pairs);
        if (rc == SBV_OK && c.profiling) c.prof_dom_used += 2 * (size_t)dom_pairs;
    }
    if (hipEventRecord(c.busy, stream) == hipSuccess) c.busy_valid = true;       // the scratch stays ordered behind whatever was enqueued
    return rc;
}

extern "C" int sbv_secp256k1_verify_batch(const uint8_t* tuples, size_t n, uint8_t* accept_bitmap) {
    SBV_ENTER(c);
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!tuples || !accept_bitmap) { g_err = "null pointer"; return SBV_EINVAL; }
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    int rc = ensure_capacity(c, n < kMaxChunk ? n : kMaxChunk);
    if (rc != SBV_OK) return rc;
    if ((rc = ensure_k256_table(c)) != SBV_OK) return rc;
    if (c.busy_valid) HIP_TRY(SBV_EDEVICE, hipStreamWaitEvent(c.stream, c.busy, 0));
    sbv_timing tm{};
    tm.n = n;
    for (size_t off = 0; off < n; off += kMaxChunk) {
        const size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[0], c.stream));
        HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.d_tuples, tuples + off * 160, m * 160, hipMemcpyHostToDevice, c.stream));
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[1], c.stream));
        if ((rc = enqueue_k256(c, c.d_tuples, m, c.d_bitmap, c.stream)) != SBV_OK) return rc;
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[3], c.stream));
        HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.h_bitmap, c.d_bitmap, (m + 7) / 8, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[4], c.stream));
        HIP_TRY(SBV_EDEVICE, hipStreamSynchronize(c.stream));
        memcpy(accept_bitmap + off / 8, c.h_bitmap, (m + 7) / 8);
        tm.h2d_us += 1e3 * ms_between(c.ev[0], c.ev[1]);
        tm.verify_us += 1e3 * ms_between(c.ev[1], c.ev[3]);
        tm.d2h_us += 1e3 * ms_between(c.ev[3], c.ev[4]);
    }
    c.busy_valid = false;
    tm.total_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    c.timing = tm;
    return SBV_OK;
}

namespace {
template <typename T>
int grow(T*& ptr, size_t& cap, size_t want_elems) {
    if (want_elems <= cap) return SBV_OK;
    size_t ncap = cap ? cap : 4096;
    while (ncap < want_elems) ncap *= 2;
    T* np = nullptr;
    HIP_TRY(SBV_ENOMEM, hipMalloc(&np, ncap * sizeof(T)));
    if (ptr) { HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize()); (void)hipFree(ptr); }
    ptr = np;
    cap = ncap;
    return SBV_OK;
}
}  // namespace

extern "C" int sbv_p256_verify_msgs_keyed(const uint8_t* msgs, const uint64_t* msg_offsets, const uint8_t* sigs,
                                          const uint64_t* sig_offsets, const uint32_t* slots, size_t n, uint8_t* accept_bitmap) {
    SBV_ENTER(c);
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!msg_offsets || !sig_offsets || !slots || !accept_bitmap) { g_err = "null pointer"; return SBV_EINVAL; }
    if (c.nkeys == 0) { g_err = "no keys registered"; return SBV_EINVAL; }
    if (n > kMaxChunk) { g_err = "batch larger than 2^21: split it"; return SBV_EINVAL; }
    // the device dereferences the offset tables: they must start at 0 and never decrease
    if (msg_offsets[0] != 0 || sig_offsets[0] != 0) { g_err = "offset tables must start at 0"; return SBV_EINVAL; }
    for (size_t i = 0; i < n; ++i)
        if (msg_offsets[i + 1] < msg_offsets[i] || sig_offsets[i + 1] < sig_offsets[i]) { g_err = "offset table is not monotone"; return SBV_EINVAL; }
    const size_t mbytes = (size_t)msg_offsets[n], sbytes = (size_t)sig_offsets[n];
    if ((mbytes && !msgs) || (sbytes && !sigs)) { g_err = "null pointer"; return SBV_EINVAL; }
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    int rc = ensure_capacity(c, n);
    if (rc != SBV_OK) return rc;
    if ((rc = grow(c.d_msgs, c.msgs_cap, mbytes + 16)) != SBV_OK) return rc;
    if ((rc = grow(c.d_sigs, c.sigs_cap, sbytes + 16)) != SBV_OK) return rc;
    if ((rc = grow(c.d_moff, c.moff_cap, n + 1)) != SBV_OK) return rc;
    if ((rc = grow(c.d_soff, c.soff_cap, n + 1)) != SBV_OK) return rc;
    if (c.busy_valid) HIP_TRY(SBV_EDEVICE, hipStreamWaitEvent(c.stream, c.busy, 0));
    sbv_timing tm{};
    tm.n = n;
    HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[0], c.stream));
    if (mbytes) HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.d_msgs, msgs, mbytes, hipMemcpyHostToDevice, c.stream));
    if (sbytes) HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.d_sigs, sigs, sbytes, hipMemcpyHostToDevice, c.stream));
    HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.d_moff, msg_offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c.stream));
    HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.d_soff, sig_offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c.stream));
    HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.d_slots, slots, n * sizeof(u32), hipMemcpyHostToDevice, c.stream));
    HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[1], c.stream));
    HIP_TRY(SBV_EDEVICE, sbv::launch_msg_frontend(c.d_msgs, c.d_moff, c.d_sigs, c.d_soff, n, reinterpret_cast<u32*>(c.d_tuples), c.stream, 0, 0, mbytes, sbytes));
    rc = enqueue_keyed(c, c.d_tuples, c.d_slots, n, c.d_bitmap, c.stream, c.ev[2]);
    if (rc != SBV_OK) return rc;
    HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[3], c.stream));
    HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.h_bitmap, c.d_bitmap, (n + 7) / 8, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[4], c.stream));
    HIP_TRY(SBV_EDEVICE, hipStreamSynchronize(c.stream));
    memcpy(accept_bitmap, c.h_bitmap, (n + 7) / 8);
    tm.h2d_us = 1e3 * ms_between(c.ev[0], c.ev[1]);
    tm.prep_us = 1e3 * ms_between(c.ev[1], c.ev[2]);       // front end + stage A
    tm.verify_us = 1e3 * ms_between(c.ev[2], c.ev[3]);
    tm.d2h_us = 1e3 * ms_between(c.ev[3], c.ev[4]);
    c.busy_valid = false;
    tm.total_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    c.timing = tm;
    return SBV_OK;
}

extern "C" int sbv_ed25519_verify_msgs(const uint8_t* sigs, const uint8_t* pks, const uint8_t* msgs, const uint64_t* msg_offsets,
                                       size_t n, uint8_t* accept_bitmap) {
    SBV_ENTER(c);
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!sigs || !pks || !msg_offsets || !accept_bitmap) { g_err = "null pointer"; return SBV_EINVAL; }
    if (n > kMaxChunk) { g_err = "batch larger than 2^21: split it"; return SBV_EINVAL; }
    if (msg_offsets[0] != 0) { g_err = "offset tables must start at 0"; return SBV_EINVAL; }
    for (size_t i = 0; i < n; ++i)
        if (msg_offsets[i + 1] < msg_offsets[i]) { g_err = "offset table is not monotone"; return SBV_EINVAL; }
    const size_t mbytes = (size_t)msg_offsets[n];
    if (mbytes && !msgs) { g_err = "null pointer"; return SBV_EINVAL; }
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    int rc = ensure_capacity(c, n);
    if (rc != SBV_OK) return rc;
    if ((rc = ensure_ed_table(c)) != SBV_OK) return rc;
    // staging: messages in d_msgs, signatures (64 B each) followed by keys (32 B each) in d_sigs, offsets in d_moff
    if ((rc = grow(c.d_msgs, c.msgs_cap, mbytes + 16)) != SBV_OK) return rc;
    if ((rc = grow(c.d_sigs, c.sigs_cap, n * 96 + 16)) != SBV_OK) return rc;
    if ((rc = grow(c.d_moff, c.moff_cap, n + 1)) != SBV_OK) return rc;
    if (c.busy_valid) HIP_TRY(SBV_EDEVICE, hipStreamWaitEvent(c.stream, c.busy, 0));
    sbv_timing tm{};
    tm.n = n;
    HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[0], c.stream));
    if (mbytes) HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.d_msgs, msgs, mbytes, hipMemcpyHostToDevice, c.stream));
    HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.d_sigs, sigs, n * 64, hipMemcpyHostToDevice, c.stream));
    HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.d_sigs + n * 64, pks, n * 32, hipMemcpyHostToDevice, c.stream));
    HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.d_moff, msg_offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c.stream));
    HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[1], c.stream));
    HIP_TRY(SBV_EDEVICE, sbv::launch_ed_msg_frontend(c.d_sigs, c.d_sigs + n * 64, c.d_msgs, c.d_moff, n,
                                                     reinterpret_cast<u32*>(c.d_tuples), c.stream));
    HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[2], c.stream));
    if ((rc = enqueue_ed25519(c, c.d_tuples, n, c.d_bitmap, c.stream)) != SBV_OK) return rc;
    HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[3], c.stream));
    HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.h_bitmap, c.d_bitmap, (n + 7) / 8, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[4], c.stream));
    HIP_TRY(SBV_EDEVICE, hipStreamSynchronize(c.stream));
    memcpy(accept_bitmap, c.h_bitmap, (n + 7) / 8);
    tm.h2d_us = 1e3 * ms_between(c.ev[0], c.ev[1]);
    tm.prep_us = 1e3 * ms_between(c.ev[1], c.ev[2]);       // the front end
    tm.verify_us = 1e3 * ms_between(c.ev[2], c.ev[3]);
    tm.d2h_us = 1e3 * ms_between(c.ev[3], c.ev[4]);
    c.busy_valid = false;
    tm.total_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    c.timing = tm;
    return SBV_OK;
}

// ---- batch signing (p256_sign.h) ----------------------------------------------------------------------
extern "C" int sbv_p256_sign_batch_dev(const void* d_keys, uint32_t n_keys, const void* d_key_index, const void* d_digests, size_t n,
                                       void* d_sigs, void* d_ok, void* hip_stream) {
    SBV_ENTER(c);
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!d_keys || !d_digests || !d_sigs || !d_ok || n_keys == 0) { g_err = "null pointer or no keys"; return SBV_EINVAL; }
    if ((reinterpret_cast<uintptr_t>(d_keys) | reinterpret_cast<uintptr_t>(d_digests) | reinterpret_cast<uintptr_t>(d_sigs) |
         reinterpret_cast<uintptr_t>(d_key_index)) & 3) {
        g_err = "misaligned device pointer";
        return SBV_EINVAL;
    }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    HIP_TRY(SBV_EDEVICE, sbv::launch_p256_sign(static_cast<const uint8_t*>(d_keys), n_keys, static_cast<const u32*>(d_key_index),
                                               static_cast<const uint8_t*>(d_digests), n, sbv::gcomb_make(c.d_g16r, c.g_bits),
                                               static_cast<uint8_t*>(d_sigs), static_cast<uint8_t*>(d_ok),
                                               static_cast<hipStream_t>(hip_stream)));
    return SBV_OK;
}

extern "C" int sbv_p256_sign_batch(const uint8_t* keys, uint32_t n_keys, const uint32_t* key_index, const uint8_t* digests, size_t n,
                                   uint8_t* sigs, uint8_t* ok) {
    SBV_ENTER(c);
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!keys || !digests || !sigs || !ok || n_keys == 0) { g_err = "null pointer or no keys"; return SBV_EINVAL; }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    // not a hot path: buffers of the call's own size, released before returning (private keys do not linger in a pool)
    uint8_t *d_keys = nullptr, *d_dig = nullptr, *d_sig = nullptr, *d_ok = nullptr;
    u32* d_idx = nullptr;
    int rc = SBV_OK;
    auto fail = [&](int code, hipError_t e) { if (e != hipSuccess && rc == SBV_OK) { g_err = hipGetErrorString(e); rc = code; } return e != hipSuccess; };
    do {
        if (fail(SBV_ENOMEM, hipMalloc(&d_keys, (size_t)n_keys * 32))) break;
        if (fail(SBV_ENOMEM, hipMalloc(&d_dig, n * 32))) break;
        if (fail(SBV_ENOMEM, hipMalloc(&d_sig, n * 64))) break;
        if (fail(SBV_ENOMEM, hipMalloc(&d_ok, n))) break;
        if (key_index && fail(SBV_ENOMEM, hipMalloc(&d_idx, n * sizeof(u32)))) break;
        if (fail(SBV_EDEVICE, hipMemcpyAsync(d_keys, keys, (size_t)n_keys * 32, hipMemcpyHostToDevice, c.stream))) break;
        if (fail(SBV_EDEVICE, hipMemcpyAsync(d_dig, digests, n * 32, hipMemcpyHostToDevice, c.stream))) break;
        if (key_index && fail(SBV_EDEVICE, hipMemcpyAsync(d_idx, key_index, n * sizeof(u32), hipMemcpyHostToDevice, c.stream))) break;
        if (fail(SBV_EDEVICE, sbv::launch_p256_sign(d_keys, n_keys, d_idx, d_dig, n, sbv::gcomb_make(c.d_g16r, c.g_bits), d_sig, d_ok, c.stream))) break;
        if (fail(SBV_EDEVICE, hipMemcpyAsync(sigs, d_sig, n * 64, hipMemcpyDeviceToHost, c.stream))) break;
        if (fail(SBV_EDEVICE, hipMemcpyAsync(ok, d_ok, n, hipMemcpyDeviceToHost, c.stream))) break;
        if (fail(SBV_EDEVICE, hipStreamSynchronize(c.stream))) break;
        (void)hipMemsetAsync(d_keys, 0, (size_t)n_keys * 32, c.stream);
        (void)hipStreamSynchronize(c.stream);
    } while (false);
    if (d_keys) (void)hipFree(d_keys);
    if (d_dig) (void)hipFree(d_dig);
    if (d_sig) (void)hipFree(d_sig);
    if (d_ok) (void)hipFree(d_ok);
    if (d_idx) (void)hipFree(d_idx);
    return rc;
}

extern "C" void* sbv_host_alloc(size_t bytes) {
    SBV_ENTER(c);
    if (!c.ready || bytes == 0) return nullptr;
    if (hipSetDevice(c.hip_dev) != hipSuccess) return nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        g_err = "hipHostMalloc failed";
        return nullptr;
    }
    return p;
}

extern "C" void sbv_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

extern "C" int sbv_p256_set_grouping(int enabled, size_t min_batch, uint32_t min_count, uint32_t max_groups) {
    Settings st;
    {
        std::lock_guard<std::mutex> lk(g_set_mu);
        g_settings.group_enabled = enabled != 0;
        if (min_batch == SBV_GROUP_MIN_BATCH_DEFAULT) {          // back to the built-in thresholds (one per scheme and cache state)
            const Settings d;
            g_settings.group_min_batch = d.group_min_batch; g_settings.group_min_batch_cold = d.group_min_batch_cold;
            g_settings.group_min_batch_ed = d.group_min_batch_ed; g_settings.group_min_batch_k256 = d.group_min_batch_k256;
            g_settings.group_min_count = d.group_min_count;          // ... and the per-scheme default thresholds and the group capacity
            g_settings.group_max = d.group_max;                      //     (a non-zero min_count / max_groups below still applies)
        } else if (min_batch) g_settings.group_min_batch = g_settings.group_min_batch_cold = g_settings.group_min_batch_ed = g_settings.group_min_batch_k256 = min_batch;
        if (min_count) g_settings.group_min_count = min_count;
        if (max_groups) g_settings.group_max = max_groups;
        st = g_settings;
    }
    for (Context* cp : live_contexts()) {
        std::lock_guard<std::mutex> lk(cp->mu);
        cp->group_enabled = st.group_enabled; cp->group_min_batch = st.group_min_batch; cp->group_min_batch_cold = st.group_min_batch_cold; cp->group_min_batch_ed = st.group_min_batch_ed; cp->group_min_batch_k256 = st.group_min_batch_k256;
        cp->group_min_count = st.group_min_count; cp->group_max = st.group_max;
    }
    return SBV_OK;
}

namespace {
sbv::KeyCache* scheme_cache(Context& c, int scheme) {
    return scheme == SBV_SCHEME_P256 ? &c.grp.kc : scheme == SBV_SCHEME_SECP256K1 ? &c.k256pool.kc : &c.edgrp.kc;
}
}  // namespace

extern "C" int sbv_key_cache(int scheme, int enabled, uint32_t capacity) {
    if (scheme < 0 || scheme > 2) return SBV_EINVAL;
    Settings st;
    {
        std::lock_guard<std::mutex> lk(g_set_mu);
        g_settings.kc_on[scheme] = enabled != 0;
        if (capacity) g_settings.kc_caps[scheme] = capacity;
        st = g_settings;
    }
    int rc = SBV_OK;
    for (Context* cp : live_contexts()) {
        std::lock_guard<std::mutex> lk(cp->mu);
        Context& c = *cp;
        c.kc_on[scheme] = st.kc_on[scheme];
        c.kc_caps[scheme] = st.kc_caps[scheme];     // a new capacity takes effect (and empties the cache) at the next grouped batch
        sbv::KeyCache& kc = *scheme_cache(c, scheme);
        if (c.ready && kc.ht) {
            hipError_t e = hipSetDevice(c.hip_dev);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            kc.enabled = c.kc_on[scheme] ? 1u : 0u;
            if (e == hipSuccess && !c.kc_on[scheme]) e = key_cache_forget(kc);   // switching it off forgets everything: the next "on" starts cold
            if (e == hipSuccess && !c.kc_on[scheme] && scheme == SBV_SCHEME_P256) e = hot_forget(c.grp);
            if (e == hipSuccess && !c.kc_on[scheme] && scheme == SBV_SCHEME_ED25519) e = ed_hot_forget(c.edgrp);
            if (e != hipSuccess) rc = fail(SBV_EDEVICE, "sbv_key_cache", e);
        }
    }
    return rc;
}
extern "C" int sbv_p256_key_cache(int enabled, uint32_t capacity) { return sbv_key_cache(SBV_SCHEME_P256, enabled, capacity); }

extern "C" int sbv_key_cache_stats(int scheme, uint32_t out[4]) {
    if (scheme < 0 || scheme > 2) return SBV_EINVAL;
    SBV_ENTER(c);
    if (!c.ready) return SBV_ENOTINIT;
    if (!out) return SBV_EINVAL;
    out[0] = out[1] = out[2] = 0;
    out[3] = c.kc_caps[scheme];
    const sbv::KeyCache& kc = *scheme_cache(c, scheme);
    if (!kc.count) return SBV_OK;
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    uint32_t h[3];
    HIP_TRY(SBV_EDEVICE, hipMemcpy(h, kc.count, sizeof h, hipMemcpyDeviceToHost));
    out[0] = h[0] < kc.cap ? h[0] : kc.cap;
    out[1] = h[1];
    out[2] = h[2];
    return SBV_OK;
}
extern "C" int sbv_p256_key_cache_stats(uint32_t out[4]) { return sbv_key_cache_stats(SBV_SCHEME_P256, out); }

extern "C" int sbv_profile_enable(int on) {
    const int level = on == 2 ? 2 : (on != 0 ? 1 : 0);
    {
        std::lock_guard<std::mutex> lk(g_set_mu);
        g_settings.profiling = level;
    }
    for (Context* cp : live_contexts()) {
        std::lock_guard<std::mutex> lk(cp->mu);
        cp->profiling = level;
    }
    return SBV_OK;
}

extern "C" int sbv_profile_read_dominant(double* dominant_us, uint64_t* dominant_launches) {
    SBV_ENTER(c);
    if (!c.ready) return SBV_ENOTINIT;
    double d = 0;
    for (size_t i = 0; i + 2 <= c.prof_dom_used; i += 2) {
        HIP_TRY(SBV_EDEVICE, hipEventSynchronize(c.prof_dom[i + 1]));
        d += 1e3 * ms_between(c.prof_dom[i], c.prof_dom[i + 1]);
    }
    if (dominant_us) *dominant_us = d;
    if (dominant_launches) *dominant_launches = c.prof_dom_used / 2;
    c.prof_dom_used = 0;
    return SBV_OK;
}

extern "C" int sbv_p256_last_group_stats(uint32_t out[4]) {
    SBV_ENTER(c);
    if (!c.ready) return SBV_ENOTINIT;
    if (!out) return SBV_EINVAL;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!c.grp.counters) return SBV_OK;
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    uint32_t h[4];
    HIP_TRY(SBV_EDEVICE, hipMemcpy(h, c.grp.counters, sizeof h, hipMemcpyDeviceToHost));
    out[0] = h[0] < c.grp.max_groups ? h[0] : c.grp.max_groups;
    out[1] = h[1];
    out[2] = h[2];
    out[3] = h[3];
    return SBV_OK;
}

extern "C" int sbv_p256_hot_keys(uint32_t max_keys, uint32_t min_hits) {
    if (max_keys > 4096) return SBV_EINVAL;
    {
        std::lock_guard<std::mutex> lk(g_set_mu);
        g_settings.hot_keys = max_keys;
        if (min_hits) g_settings.hot_min_hits = min_hits;
    }
    int rc = SBV_OK;
    for (Context* cp : live_contexts()) {
        std::lock_guard<std::mutex> lk(cp->mu);
        Context& c = *cp;
        if (min_hits) c.hot_min_hits = min_hits;
        if (c.hot_keys == max_keys) continue;
        c.hot_keys = max_keys;
        if (!c.ready || !c.grp.ktab) continue;
        // another pool size: the comb pools are rebuilt (with the cache) by the next grouped batch
        hipError_t e = hipSetDevice(c.hip_dev);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { rc = fail(SBV_EDEVICE, "sbv_p256_hot_keys", e); continue; }
        free_group_buffers(c);
    }
    return rc;
}

extern "C" int sbv_p256_hot_key_stats(uint32_t out[4]) {
    SBV_ENTER(c);
    if (!c.ready) return SBV_ENOTINIT;
    if (!out) return SBV_EINVAL;
    out[0] = out[2] = 0;
    out[1] = c.grp.wide_cap;
    out[3] = c.hot_min_hits;
    if (!c.grp.hot) return SBV_OK;
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    uint32_t h[4];
    HIP_TRY(SBV_EDEVICE, hipMemcpy(h, c.grp.hot, sizeof h, hipMemcpyDeviceToHost));
    out[0] = h[0] < c.grp.wide_cap ? h[0] : c.grp.wide_cap;
    out[2] = h[2];
    return SBV_OK;
}

// Hot keys of the Ed25519 scheme (ed25519_group.h): the pool of 16-bit combs of -A
extern "C" int sbv_ed25519_hot_keys(uint32_t max_keys, uint32_t min_hits) {
    if (max_keys > 4096) return SBV_EINVAL;
    {
        std::lock_guard<std::mutex> lk(g_set_mu);
        g_settings.ed_hot_keys = max_keys;
        if (min_hits) g_settings.ed_hot_min_hits = min_hits;
    }
    int rc = SBV_OK;
    for (Context* cp : live_contexts()) {
        std::lock_guard<std::mutex> lk(cp->mu);
        Context& c = *cp;
        if (min_hits) c.ed_hot_min_hits = min_hits;
        if (c.ed_hot_keys == max_keys) continue;
        c.ed_hot_keys = max_keys;
        if (!c.ready || !c.edgrp.ktab) continue;
        // another pool size: this scheme's comb pool (with its cache) is rebuilt by the next grouped batch
        hipError_t e = hipSetDevice(c.hip_dev);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { rc = fail(SBV_EDEVICE, "sbv_ed25519_hot_keys", e); continue; }
        sbv::EdGroupBuffers& eb = c.edgrp;
        if (eb.ktab) (void)hipFree(eb.ktab);
        if (eb.okb) (void)hipFree(eb.okb);
        if (eb.ungxy) (void)hipFree(eb.ungxy);
        if (eb.kvalid) (void)hipFree(eb.kvalid);
        key_cache_free(eb.kc);
        ed_hot_free(eb);
        eb = sbv::EdGroupBuffers();
    }
    return rc;
}

extern "C" int sbv_ed25519_hot_key_stats(uint32_t out[4]) {
    SBV_ENTER(c);
    if (!c.ready) return SBV_ENOTINIT;
    if (!out) return SBV_EINVAL;
    out[0] = out[2] = 0;
    out[1] = c.edgrp.wide_cap;
    out[3] = c.ed_hot_min_hits;
    if (!c.edgrp.hot) return SBV_OK;
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    uint32_t h[4];
    HIP_TRY(SBV_EDEVICE, hipMemcpy(h, c.edgrp.hot, sizeof h, hipMemcpyDeviceToHost));
    out[0] = h[0] < c.edgrp.wide_cap ? h[0] : c.edgrp.wide_cap;
    out[2] = h[2];
    return SBV_OK;
}

// 1 = promoted comb `index` equals the host builder's comb of -A for its key, entry by entry (canonical affine-Niels entries)
extern "C" int sbv_ed25519_hot_selfcheck(uint32_t index) {
    SBV_ENTER(c);
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    sbv::EdGroupBuffers& b = c.edgrp;
    if (!b.wtab || !b.kwide || index >= b.wide_cap) { g_err = "no such comb in the Ed25519 hot-key pool"; return SBV_EINVAL; }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    std::vector<u32> kw(b.kc.cap);
    HIP_TRY(SBV_EDEVICE, hipMemcpy(kw.data(), b.kwide, kw.size() * sizeof(u32), hipMemcpyDeviceToHost));
    size_t slot = kw.size();
    for (size_t i = 0; i < kw.size(); ++i) if (kw[i] == index) slot = i;
    if (slot == kw.size()) { g_err = "sbv_ed25519_hot_selfcheck: no promoted key has this index"; return SBV_EINVAL; }
    u32 key[8];
    HIP_TRY(SBV_EDEVICE, hipMemcpy(key, b.kc.keys + slot * 16, 32, hipMemcpyDeviceToHost));
    sbv::ept A;
    if (!sbv::ed_decompress(A, key)) return 0;                          // only valid keys are promoted
    sbv::fe25_neg(A.X, A.X);
    sbv::fe25_neg(A.T, A.T);
    std::vector<sbv::aniels> want((size_t)SBV_ED_HOT_WINDOWS * SBV_ED_HOT_PER_WINDOW);
    {
        std::vector<std::thread> th;
        for (int j = 0; j < SBV_ED_HOT_WINDOWS; ++j)
            th.emplace_back([&, j] { sbv::build_ed_window_of(A, SBV_ED_HOT_BITS, j, want.data() + (size_t)j * SBV_ED_HOT_PER_WINDOW); });
        for (auto& t : th) t.join();
    }
    std::vector<uint8_t> got(SBV_ED_HOT_COMB_BYTES);
    HIP_TRY(SBV_EDEVICE, hipMemcpy(got.data(), b.wtab + (size_t)index * SBV_ED_HOT_COMB_BYTES, got.size(), hipMemcpyDeviceToHost));
    for (size_t e = 0; e < want.size(); ++e)
        if (memcmp(got.data() + e * SBV_ED_HOT_PITCH, &want[e], sizeof(sbv::aniels)) != 0) return 0;
    return 1;
}

// Diagnostics (round 6): every promoted comb of context `device` against the host builder, and the consistency of kwide / wowner.
// out[0] = promoted slots, out[1] = combs that differ, out[2] = first differing comb's index, out[3] = its first differing entry,
// out[4] = number of differing entries in it, out[5] = kwide / wowner inconsistencies, out[6] = slots whose kwide index is shared with
// another slot, out[7] = hot[0].  Slow (a host build per comb); tests and tools only.
extern "C" int sbv_debug_hot_check(int device, uint32_t out[8]) {
    Context* cp;
    { std::lock_guard<std::mutex> lk(g_mu); cp = context_of(device, false); }
    if (!cp || !out) return SBV_EINVAL;
    std::lock_guard<std::mutex> lkc(cp->mu);
    Context& c = *cp;
    if (!c.ready) return SBV_ENOTINIT;
    sbv::GroupBuffers& b = c.grp;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    if (!b.wtab || !b.kwide) return SBV_OK;
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    std::vector<u32> kw(b.kc.cap), wo(b.wide_cap), hot(4);
    HIP_TRY(SBV_EDEVICE, hipMemcpy(kw.data(), b.kwide, kw.size() * sizeof(u32), hipMemcpyDeviceToHost));
    HIP_TRY(SBV_EDEVICE, hipMemcpy(wo.data(), b.wowner, wo.size() * sizeof(u32), hipMemcpyDeviceToHost));
    HIP_TRY(SBV_EDEVICE, hipMemcpy(hot.data(), b.hot, 4 * sizeof(u32), hipMemcpyDeviceToHost));
    out[7] = hot[0];
    if (getenv("SBV_DEBUG_HOT_DUMP")) {          // tools/stress_logical.py: the whole bookkeeping of the context on stderr
        u32 cnt[4] = {0, 0, 0, 0};
        (void)hipMemcpy(cnt, b.kc.count, sizeof cnt, hipMemcpyDeviceToHost);
        const u32 entries = cnt[0] < b.kc.cap ? cnt[0] : b.kc.cap;
        std::vector<u32> keys((size_t)entries * 16), kh(b.kc.cap);
        std::vector<uint8_t> kv(entries), kf(entries);
        (void)hipMemcpy(keys.data(), b.kc.keys, keys.size() * sizeof(u32), hipMemcpyDeviceToHost);
        (void)hipMemcpy(kh.data(), b.khits, kh.size() * sizeof(u32), hipMemcpyDeviceToHost);
        (void)hipMemcpy(kv.data(), b.kvalid, entries, hipMemcpyDeviceToHost);
        (void)hipMemcpy(kf.data(), b.kfull, entries, hipMemcpyDeviceToHost);
        fprintf(stderr, "[hot dump ctx %d] cache entries %u (hits %u misses %u) hot %u %u %u %u tick %u\n", c.device, cnt[0], cnt[1], cnt[2], hot[0], hot[1], hot[2], hot[3], b.hot_tick);
        std::unordered_map<std::string, u32> seen;
        for (u32 sl = 0; sl < entries; ++sl) {
            const std::string k((const char*)&keys[(size_t)sl * 16], 64);
            const auto it = seen.find(k);
            const int dup = it == seen.end() ? -1 : (int)it->second;
            if (it == seen.end()) seen.emplace(k, sl);
            if (kh[sl] || kw[sl] != 0xFFFFFFFFu || dup >= 0)
                fprintf(stderr, "  slot %u key %08x%08x valid %u full %u hits %u wide %d dup_of %d\n", sl, keys[(size_t)sl * 16], keys[(size_t)sl * 16 + 1], kv[sl], kf[sl], kh[sl], (int)kw[sl], dup);
        }
    }
    const size_t stride = sbv::gcomb_entries(SBV_HOT_BITS), per = (size_t)1 << (SBV_HOT_BITS - 1);
    const size_t W = stride / per;
    std::vector<sbv::apt> want(stride), got(stride);
    std::vector<u32> users(b.wide_cap, 0);
    unsigned hw = std::thread::hardware_concurrency();
    for (size_t slot = 0; slot < kw.size(); ++slot) {
        const u32 w = kw[slot];
        if (w == 0xFFFFFFFFu) continue;
        ++out[0];
        if (w >= b.wide_cap) { ++out[5]; continue; }
        if (wo[w] != slot) ++out[5];
        if (users[w]++) ++out[6];
        uint8_t key[64];
        HIP_TRY(SBV_EDEVICE, hipMemcpy(key, b.kc.keys + slot * 16, 64, hipMemcpyDeviceToHost));
        if (!sbv::host_build_wide_key_table(key, SBV_HOT_BITS, want.data(), (int)(hw > 32 ? 32 : (hw ? hw : 1)))) continue;
        HIP_TRY(SBV_EDEVICE, hipMemcpy(got.data(), b.wtab + (size_t)w * stride, stride * sizeof(sbv::apt), hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0;
        const size_t lim = (W - 1) * per + (((size_t)1 << (SBV_HOT_BITS / 2)) - 1);
        for (size_t e = 0; e < lim; ++e)
            if (memcmp(&got[e], &want[e], sizeof(sbv::apt)) != 0) { if (!bad) first = e; ++bad; }
        if (bad) { if (!out[1]) { out[2] = w; out[3] = (u32)first; out[4] = (u32)bad; } ++out[1]; }
    }
    return SBV_OK;
}

// Pools the default device's grouped P-256 step really holds (they exist after its first grouped batch): see include/sbv.h
extern "C" int sbv_p256_pool_stats(uint32_t out[6]) {
    SBV_ENTER(c);
    if (!c.ready) return SBV_ENOTINIT;
    if (!out) return SBV_EINVAL;
    out[0] = c.grp.ktab ? c.grp.kc.cap : 0;
    out[1] = c.grp.ktab ? c.grp.max_groups : 0;
    out[2] = c.pools_shrunk ? 1u : 0u;
    out[3] = c.group_nomem_events;
    out[4] = c.grp.wide_cap;
    out[5] = (u32)c.mem_share;
    return SBV_OK;
}

// Diagnostics: is promoted comb number `index` what the host builder produces for its key?  Windows 0..15 byte for byte; of the top
// window (the carry of the signed recoding) the babies, of which entry 1 is the only one ever read.  1 = equal, 0 = different.
extern "C" int sbv_p256_hot_selfcheck(uint32_t index) {
    SBV_ENTER(c);
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    sbv::GroupBuffers& b = c.grp;
    if (!b.wtab || !b.kwide) { g_err = "no hot-key pool"; return SBV_EINVAL; }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    std::vector<u32> kw(b.kc.cap);
    HIP_TRY(SBV_EDEVICE, hipMemcpy(kw.data(), b.kwide, kw.size() * sizeof(u32), hipMemcpyDeviceToHost));
    size_t slot = kw.size();
    for (size_t i = 0; i < kw.size(); ++i) if (kw[i] == index) slot = i;
    if (slot == kw.size()) { g_err = "sbv_p256_hot_selfcheck: no promoted key has this index"; return SBV_EINVAL; }
    uint8_t key[64];
    HIP_TRY(SBV_EDEVICE, hipMemcpy(key, b.kc.keys + slot * 16, 64, hipMemcpyDeviceToHost));
    const size_t stride = sbv::gcomb_entries(SBV_HOT_BITS), per = (size_t)1 << (SBV_HOT_BITS - 1);
    std::vector<sbv::apt> want(stride), got(stride);
    unsigned hw = std::thread::hardware_concurrency();
    if (!sbv::host_build_wide_key_table(key, SBV_HOT_BITS, want.data(), (int)(hw > 32 ? 32 : (hw ? hw : 1)))) return 0;
    HIP_TRY(SBV_EDEVICE, hipMemcpy(got.data(), b.wtab + (size_t)index * stride, stride * sizeof(sbv::apt), hipMemcpyDeviceToHost));
    const size_t W = stride / per;                                            // 17
    if (memcmp(got.data(), want.data(), (W - 1) * per * sizeof(sbv::apt)) != 0) return 0;
    return memcmp(got.data() + (W - 1) * per, want.data() + (W - 1) * per, (((size_t)1 << (SBV_HOT_BITS / 2)) - 1) * sizeof(sbv::apt)) == 0 ? 1 : 0;
}

extern "C" int sbv_p256_last_table_classes(uint32_t out[3]) {
    SBV_ENTER(c);
    if (!c.ready) return SBV_ENOTINIT;
    if (!out) return SBV_EINVAL;
    out[0] = out[1] = out[2] = 0;
    if (!c.grp.counters) return SBV_OK;
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    uint32_t h[SBV_GROUP_COUNTERS];
    HIP_TRY(SBV_EDEVICE, hipMemcpy(h, c.grp.counters, sizeof h, hipMemcpyDeviceToHost));
    out[0] = h[5]; out[1] = h[6]; out[2] = h[7];
    return SBV_OK;
}

extern "C" int sbv_profile_read(double* prep_us, double* verify_us, uint64_t* launches) {
    SBV_ENTER(c);
    if (!c.ready) return SBV_ENOTINIT;
    double p = 0, v = 0;
    for (size_t i = 0; i + 3 <= c.prof_used; i += 3) {
        HIP_TRY(SBV_EDEVICE, hipEventSynchronize(c.prof_events[i + 2]));
        p += 1e3 * ms_between(c.prof_events[i], c.prof_events[i + 1]);
        v += 1e3 * ms_between(c.prof_events[i + 1], c.prof_events[i + 2]);
    }
    if (prep_us) *prep_us = p;
    if (verify_us) *verify_us = v;
    if (launches) *launches = c.prof_used / 3;
    c.prof_used = 0;
    return SBV_OK;
}

extern "C" int sbv_last_timing(sbv_timing* out) {
    if (!out) return SBV_EINVAL;
    SBV_ENTER(c);
    *out = c.timing;
    return SBV_OK;
}

// =====================================================================================================================
// All GPUs of the node behind one process (BASELINE.json north_star: "sharded across the 8 GPUs of one node
// with an RCCL all-gather of the per-lane accept/reject bitmap over xGMI only when a batch outgrows one GPU").
//
//   * tuples are independent: a batch is split into contiguous shards, one per device, each a multiple of the granule
//     lcm(512, 8 * group) tuples — whole bitmap bytes per device and, for consenter-signature batches (group = signatures
//     per proposal, e.g. 11 at N = 16), whole proposals, so the device can also emit the per-proposal quorum bit;
//   * every device gets its shard over its own PCIe link from its own host thread, verifies it with the same grouped step
//     as the single-device entry and leaves its bitmap shard in its slot of a device-resident gather buffer;
//   * when more than one device took part, ONE in-place ncclAllGather (uint8, shard bytes per rank) over the per-device
//     streams leaves the full bitmap on every device and a single D2H returns it; RCCL is resolved with dlopen the first
//     time it is needed, so single-GPU deployments never load it;
//   * a batch below min_per_device * 2 tuples is not split: it goes whole to one device, round-robin ("replicas", no
//     collective at all).
namespace {

struct RcclApi {
    void* handle = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;     // key-affine partition: OR of disjoint bitmaps = their byte sum
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::vector<void*> comms;          // one communicator per initialised device, in g_devs order
    bool ready = false;
} g_rccl;
std::vector<int> g_devs;               // devices initialised by sbv_init_all, ascending
std::atomic<unsigned> g_rr{0};         // round-robin cursor of the replica route
// Tuples per device below which a batch is not split further.  Two values, chosen per batch by shard_min_for():
//   g_shard_min       a batch over MANY signers (client requests): every device builds the tables of every key it meets, a
//                     cost that does not shrink with the shard; measured cold steps (profiles/r03/sweep_sizes_r03u.jsonl):
//                     2^20 3.30 ms, 2^19 2.17, 2^18 1.66, 2^17 1.44, 2^16 1.35 — still falling at 2^17, flat below.
//   g_shard_min_few   a batch over a HANDFUL of signers (consenter commit signatures: 16 keys at N = 16, configs[3]; the keys
//                     are registered or sit in the key-table cache after the first piece): 2^16 tuples already run at
//                     181 M/s warm (0.36 ms), so 550 000 tuples really span 8 devices (68 750 each).
size_t g_shard_min = (size_t)1 << 17;
size_t g_shard_min_few = (size_t)1 << 16;
bool g_shard_min_env = false;           // SBV_SHARD_MIN given: it overrides both
size_t g_shard_piece = (size_t)1 << 18;  // verify_shard: tuples per upload piece when the key-table cache is on (SBV_SHARD_PIECE)
// How a batch that spans devices is partitioned: 0 = contiguous ranges of tuples (every device sees every key), 1 = by a
// hash of the public key (device g builds the tables of its keys only; every device receives the whole batch).
// g_shard_parts: parts of the key-affine partition; 0 = one per device.  More parts than devices run one after another on
// their device (that is how one GPU rehearses an 8-GPU partition: tests, bench.py's projection leg).
std::atomic<int> g_shard_mode{0};
std::atomic<unsigned> g_shard_parts{0};

void rccl_teardown() {                 // g_mu held
    if (g_rccl.ready) for (void* cm : g_rccl.comms) if (cm) (void)g_rccl.CommDestroy(cm);
    g_rccl.comms.clear();
    g_rccl.ready = false;
    g_devs.clear();
}

// g_mu held.  Failure is not fatal: the sharded entry then gathers the shards through the host.
bool rccl_setup() {
    if (g_rccl.ready) return true;
    if (!g_rccl.handle) {
        g_rccl.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!g_rccl.handle) g_rccl.handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!g_rccl.handle) return false;
        g_rccl.CommInitAll = reinterpret_cast<int (*)(void**, int, const int*)>(dlsym(g_rccl.handle, "ncclCommInitAll"));
        g_rccl.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(g_rccl.handle, "ncclCommDestroy"));
        g_rccl.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(g_rccl.handle, "ncclAllGather"));
        g_rccl.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(g_rccl.handle, "ncclAllReduce"));
        g_rccl.GroupStart = reinterpret_cast<int (*)()>(dlsym(g_rccl.handle, "ncclGroupStart"));
        g_rccl.GroupEnd = reinterpret_cast<int (*)()>(dlsym(g_rccl.handle, "ncclGroupEnd"));
        g_rccl.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(g_rccl.handle, "ncclGetErrorString"));
    }
    if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GroupStart || !g_rccl.GroupEnd) return false;
    std::vector<int> hip_devs;
    for (int d : g_devs) {
        const int h = hipdev_of(d);
        if (std::find(hip_devs.begin(), hip_devs.end(), h) != hip_devs.end()) return false;     // logical devices folded onto one GPU: a communicator needs one rank per physical device
        hip_devs.push_back(h);
    }
    g_rccl.comms.assign(g_devs.size(), nullptr);
    if (g_rccl.CommInitAll(g_rccl.comms.data(), (int)hip_devs.size(), hip_devs.data()) != 0) { g_rccl.comms.clear(); return false; }
    g_rccl.ready = true;
    return true;
}

size_t gcd_sz(size_t a, size_t b) { while (b) { const size_t t = a % b; a = b; b = t; } return a; }
size_t shard_granule(size_t group) {
    const size_t g8 = 8 * (group ? group : 1);
    return 512 / gcd_sz(512, g8) * g8;            // lcm(512, 8 * group)
}

// per proposal: at least `quorum` accepted signatures by DISTINCT public keys (>= Q distinct signers:
// internal/bft/viewchanger.go:681-727; Q from internal/bft/util.go:183-187).  One lane per proposal; bits LSB-first.
__global__ __launch_bounds__(256) void k_quorum_bits(const uint8_t* __restrict__ tuples, const uint8_t* __restrict__ bitmap,
                                                     size_t nprops, u32 group, u32 quorum, uint8_t* __restrict__ qbitmap) {
    const size_t pidx = (size_t)blockIdx.x * 256 + threadIdx.x;
    bool decided = false;
    if (pidx < nprops) {
        const size_t first = pidx * group;
        u32 count = 0;
        for (u32 i = 0; i < group; ++i) {
            const size_t t = first + i;
            if (!((bitmap[t >> 3] >> (t & 7)) & 1u)) continue;
            const uint4* ki = reinterpret_cast<const uint4*>(tuples + t * SBV_TUPLE_BYTES + 96);
            bool dup = false;
            for (u32 j = 0; j < i && !dup; ++j) {
                const size_t u = first + j;
                if (!((bitmap[u >> 3] >> (u & 7)) & 1u)) continue;
                const uint4* kj = reinterpret_cast<const uint4*>(tuples + u * SBV_TUPLE_BYTES + 96);
                bool same = true;
                for (int w = 0; w < 4; ++w) {
                    const uint4 a = ki[w], b = kj[w];
                    same = same && a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w;
                }
                dup = same;
            }
            if (!dup) ++count;
        }
        decided = count >= quorum;
    }
    const unsigned long long m = __ballot(decided);
    const int lane = threadIdx.x & 63;
    const size_t wave_first = pidx - (size_t)lane;
    if (lane < 8) {
        const size_t byte = (wave_first >> 3) + (size_t)lane;
        if (byte < ((nprops + 7) >> 3)) qbitmap[byte] = (uint8_t)(m >> (8 * lane));
    }
}

// ---- key-affine partition ---------------------------------------------------------------------------------
// A contiguous split hands every device tuples of EVERY signer, so every device builds every key's tables: the one part of
// a cold step that does not shrink with the shard.  Partitioning by a hash of the public key instead gives device g the
// tuples of "its" keys only — K / G tables and n / G tuples.  Part g of G of a device-resident batch:
//   k_part_select   membership by key hash, compaction of the member indices (one atomic per wavefront)
//   k_part_gather   the member tuples copied into a dense staging batch (16 bytes per lane, coalesced)
//   [the ordinary step on the dense batch]
//   k_part_scatter  dense verdict bit j -> bit idx[j] of the caller's bitmap (zeroed first; atomicOr on 32-bit words)
// The hash is NOT the grouping hash table's slot hash (its low bits would then be constant per device and every key would
// land in 1 / G of the table).
__device__ __forceinline__ u32 part_of_key(const uint8_t* tuples, size_t i, u32 parts) {
    const u32* k = reinterpret_cast<const u32*>(tuples + i * SBV_TUPLE_BYTES + 96);
    u32 h = 0x2545F491u;
#pragma unroll
    for (int j = 0; j < 16; ++j) { h = (h ^ k[j]) * 0x9E3779B1u; h ^= h >> 13; }
    h *= 0x85EBCA77u;
    return (h >> 11) % parts;
}
__global__ __launch_bounds__(256) void k_part_select(const uint8_t* __restrict__ tuples, size_t n, u32 part, u32 parts,
                                                     u32* __restrict__ idx, u32* __restrict__ count) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool mine = i < n && part_of_key(tuples, i, parts) == part;
    const unsigned long long m = __ballot(mine);
    const int lane = threadIdx.x & 63;
    u32 base = 0;
    if (lane == 0 && m) base = atomicAdd(count, (u32)__popcll(m));
    base = (u32)__shfl((int)base, 0, 64);
    if (mine) idx[base + (u32)__popcll(m & ((1ull << lane) - 1ull))] = (u32)i;
}
__global__ __launch_bounds__(256) void k_part_gather(const uint8_t* __restrict__ tuples, const u32* __restrict__ idx, size_t members,
                                                     uint8_t* __restrict__ dense) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;            // 16-byte element of the dense batch
    const size_t j = e / 10, part = e - j * 10;
    if (j >= members) return;
    reinterpret_cast<uint4*>(dense)[e] = reinterpret_cast<const uint4*>(tuples + (size_t)idx[j] * SBV_TUPLE_BYTES)[part];
}
__global__ __launch_bounds__(256) void k_part_scatter(const uint8_t* __restrict__ dense_bitmap, const u32* __restrict__ idx, size_t members,
                                                      u32* __restrict__ out_words) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= members) return;
    if ((dense_bitmap[j >> 3] >> (j & 7)) & 1u) { const u32 i = idx[j]; atomicOr(&out_words[i >> 5], 1u << (i & 31)); }
}

int ensure_part_buffers(Context& c, PartBuffers& pb, size_t n) {
    const size_t want = (n + 1023) & ~(size_t)1023;
    if (want <= pb.cap) return SBV_OK;
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    if (pb.d_dense) (void)hipFree(pb.d_dense);
    if (pb.d_idx) (void)hipFree(pb.d_idx);
    if (pb.d_bits) (void)hipFree(pb.d_bits);
    pb.d_dense = nullptr; pb.d_idx = nullptr; pb.d_bits = nullptr; pb.cap = 0;
    if (!pb.d_count) HIP_TRY(SBV_ENOMEM, hipMalloc(&pb.d_count, 64));
    if (!pb.h_count) HIP_TRY(SBV_ENOMEM, hipHostMalloc(&pb.h_count, 64, hipHostMallocDefault));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&pb.d_dense, want * SBV_TUPLE_BYTES));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&pb.d_idx, want * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&pb.d_bits, want / 8 + 64));
    pb.cap = want;
    (void)c;
    return SBV_OK;
}

// Part `part` of `parts` of n device-resident tuples on `stream`; c.mu held.  d_out_words: ceil(n / 32) words, receives the
// verdict bits of this part's tuples (every other bit 0).  One host round trip (the member count sizes the launches).
int part_enqueue(Context& c, const uint8_t* d_tuples, size_t n, u32 part, u32 parts, u32* d_out_words, hipStream_t stream, size_t* members_out,
                 bool zero_first = true) {
    PartBuffers& pb = g_part[c.device];
    if (zero_first) HIP_TRY(SBV_EDEVICE, hipMemsetAsync(d_out_words, 0, ((n + 31) / 32) * sizeof(u32), stream));
    size_t total = 0;
    for (size_t off = 0; off < n; off += kMaxChunk) {                  // kMaxChunk is a multiple of 32
        const size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        int rc = ensure_part_buffers(c, pb, m);
        if (rc != SBV_OK) return rc;
        const uint8_t* src = d_tuples + off * SBV_TUPLE_BYTES;
        HIP_TRY(SBV_EDEVICE, hipMemsetAsync(pb.d_count, 0, sizeof(u32), stream));
        hipLaunchKernelGGL(k_part_select, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, src, m, part, parts, pb.d_idx, pb.d_count);
        HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(pb.h_count, pb.d_count, sizeof(u32), hipMemcpyDeviceToHost, stream));
        // one host round trip: the member count sizes the launches (a form without it — launches sized for an upper bound, the
        // count read on the device — was measured in round 4 and did not move the projection: profiles/r04/projection_nosync*_r04a.json)
        HIP_TRY(SBV_EDEVICE, hipStreamSynchronize(stream));
        const size_t members = *pb.h_count;
        total += members;
        if (members == 0) continue;
        hipLaunchKernelGGL(k_part_gather, dim3((unsigned)((members * 10 + 255) / 256)), dim3(256), 0, stream, src, pb.d_idx, members, pb.d_dense);
        if ((rc = ensure_capacity(c, members)) != SBV_OK) return rc;
        if (c.busy_valid) HIP_TRY(SBV_EDEVICE, hipStreamWaitEvent(stream, c.busy, 0));
        rc = enqueue(c, pb.d_dense, members, pb.d_bits, stream, nullptr, nullptr, nullptr, nullptr, n);
        if (hipEventRecord(c.busy, stream) == hipSuccess) c.busy_valid = true;
        if (rc != SBV_OK) return rc;
        hipLaunchKernelGGL(k_part_scatter, dim3((unsigned)((members + 255) / 256)), dim3(256), 0, stream, pb.d_bits, pb.d_idx, members, d_out_words + off / 32);
        HIP_TRY(SBV_EDEVICE, hipGetLastError());
        if (off + kMaxChunk < n) HIP_TRY(SBV_EDEVICE, hipStreamSynchronize(stream));       // the staging buffers are reused by the next chunk
    }
    if (members_out) *members_out = total;
    return SBV_OK;
}

int grow_bytes(uint8_t*& ptr, size_t& cap, size_t want) {
    if (want <= cap) return SBV_OK;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr; cap = 0;
    HIP_TRY(SBV_ENOMEM, hipMalloc(&ptr, want));
    cap = want;
    return SBV_OK;
}

// One device's share of a sharded call; c.mu held.  Two upload slots and the copy stream: while the kernels of piece i run
// on c.stream, the tuples of piece i + 1 are on their way over PCIe (168 MB per 2^20 tuples: 3.3 ms at 50 GB/s, as long as
// the kernels themselves).  Pieces are 2^18 tuples (a multiple of the granule) when the key-table cache is on — the first
// piece builds the signers' combs, the others find them (0.87 ms per piece, 1.7 ms for the first; one cold 2^20 step is
// 3.45 ms) — and whole launches of up to kMaxChunk tuples when it is off (every cold piece would rebuild every table).
// The bitmap is written straight into this device's slot of the gather buffer, the quorum bits of each piece behind it;
// nothing waits on the host until the last piece is enqueued.  h2d_us: the copies' own durations, summed; kern_us: from the
// end of the first copy to the end of the last kernel (waits for later copies included).
int verify_shard(Context& c, const uint8_t* h_tuples, size_t m, size_t group, u32 quorum, uint8_t* d_slot, uint8_t* d_qslot,
                 double* h2d_us, double* kern_us) {
    if (!c.ready) { g_err = "device not initialised"; return SBV_ENOTINIT; }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    const size_t gran = shard_granule(group);
    const size_t want_piece = c.kc_on[0] && c.group_enabled ? g_shard_piece : kMaxChunk;
    size_t chunk = (want_piece < kMaxChunk ? want_piece : kMaxChunk) / gran * gran;
    if (chunk == 0) chunk = kMaxChunk / gran * gran;
    if (chunk == 0) { g_err = "group too large"; return SBV_EINVAL; }
    size_t pieces = (m + chunk - 1) / chunk;
    if (pieces > 1) {
        // EQUAL pieces (round 6): 550 000 signatures in pieces of 2^18 used to end with a piece of 32 000 that paid the step's fixed
        // chain (grouping, sort, classes: ~0.4 ms) for a tenth of the work, behind the last upload
        const size_t even = ((m + pieces - 1) / pieces + gran - 1) / gran * gran;
        if (even && even <= chunk) { chunk = even; pieces = (m + chunk - 1) / chunk; }
    }
    int rc = ensure_capacity(c, m < chunk ? m : chunk);
    if (rc != SBV_OK) return rc;
    ShardBuffers& sbuf = g_shard[c.device];
    if (pieces > 1) {
        rc = grow_bytes(sbuf.d_stage2, sbuf.stage2_cap, chunk * SBV_TUPLE_BYTES);
        if (rc != SBV_OK) return rc;
        if (!c.copy_stream && hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking) != hipSuccess) { g_err = "hipStreamCreate failed"; return SBV_EDEVICE; }
    }
    hipStream_t up = pieces > 1 ? c.copy_stream : c.stream;
    uint8_t* stage[2] = {c.d_tuples, pieces > 1 ? sbuf.d_stage2 : c.d_tuples};
    // events: per piece copy start / copy end / kernels end (timing), destroyed before returning
    std::vector<hipEvent_t> ev(3 * pieces, nullptr);
    auto drop_events = [&] { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); };
    for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) { drop_events(); g_err = "hipEventCreate failed"; return SBV_EDEVICE; }
    hipError_t he = hipSuccess;
    auto step = [&](hipError_t r) { if (he == hipSuccess) he = r; };
    if (c.busy_valid) {                        // kernels of an earlier call (pipelined host entry, a key-affine part) may still be using the scratch and group buffers
        step(hipStreamWaitEvent(c.stream, c.busy, 0));
        if (pieces > 1) step(hipStreamWaitEvent(up, c.busy, 0));
    }
    size_t i = 0;
    for (size_t off = 0; off < m && rc == SBV_OK && he == hipSuccess; off += chunk, ++i) {
        const size_t k = m - off < chunk ? m - off : chunk;
        uint8_t* d_in = stage[i & 1];
        if (i >= 2) step(hipStreamWaitEvent(up, ev[3 * (i - 2) + 2], 0));       // the slot's previous kernels are done with it
        step(hipEventRecord(ev[3 * i], up));
        step(hipMemcpyAsync(d_in, h_tuples + off * SBV_TUPLE_BYTES, k * SBV_TUPLE_BYTES, hipMemcpyHostToDevice, up));
        step(hipEventRecord(ev[3 * i + 1], up));
        if (pieces > 1) step(hipStreamWaitEvent(c.stream, ev[3 * i + 1], 0));
        if (he != hipSuccess) break;
        rc = enqueue(c, d_in, k, d_slot + off / 8, c.stream, nullptr);
        if (rc != SBV_OK) break;
        if (d_qslot && group > 0 && quorum > 0) {
            const size_t props = k / group;                 // k is a multiple of group except for a ragged tail, which gets no bit
            if (props)
                hipLaunchKernelGGL(k_quorum_bits, dim3((unsigned)((props + 255) / 256)), dim3(256), 0, c.stream, d_in, d_slot + off / 8,
                                   props, (u32)group, quorum, d_qslot + (off / group) / 8);
        }
        step(hipEventRecord(ev[3 * i + 2], c.stream));
    }
    // drain in every case: the slots and the caller's host buffer must not be in use when this returns
    if (pieces > 1) step(hipStreamSynchronize(up));
    step(hipStreamSynchronize(c.stream));
    if (rc == SBV_OK && he != hipSuccess) rc = fail(SBV_EDEVICE, "verify_shard", he);
    if (rc == SBV_OK) {
        if (h2d_us) for (size_t j = 0; j < pieces; ++j) *h2d_us += 1e3 * ms_between(ev[3 * j], ev[3 * j + 1]);
        if (kern_us) *kern_us += 1e3 * ms_between(ev[1], ev[3 * (pieces - 1) + 2]);
    }
    drop_events();
    c.busy_valid = false;
    return rc;
}

// The registered-key form of k_quorum_bits: signature t of a proposal carries a key SLOT (equal keys share a slot: sbv_p256_register_keys),
// so ">= quorum accepted signatures by distinct signers" (internal/bft/viewchanger.go:681-727) counts distinct accepted slots.
__global__ __launch_bounds__(256) void k_quorum_bits_slots(const u32* __restrict__ slots, const uint8_t* __restrict__ bitmap,
                                                           size_t nprops, u32 group, u32 quorum, uint8_t* __restrict__ qbitmap) {
    const size_t pidx = (size_t)blockIdx.x * 256 + threadIdx.x;
    bool decided = false;
    if (pidx < nprops) {
        const size_t first = pidx * group;
        u32 count = 0;
        for (u32 i = 0; i < group; ++i) {
            const size_t t = first + i;
            if (!((bitmap[t >> 3] >> (t & 7)) & 1u)) continue;
            const u32 si = slots[t];
            bool dup = false;
            for (u32 j = 0; j < i && !dup; ++j) {
                const size_t u = first + j;
                dup = ((bitmap[u >> 3] >> (u & 7)) & 1u) && slots[u] == si;
            }
            if (!dup) ++count;
        }
        decided = count >= quorum;
    }
    const unsigned long long m = __ballot(decided);
    const int lane = threadIdx.x & 63;
    const size_t wave_first = pidx - (size_t)lane;
    if (lane < 8) {
        const size_t byte = (wave_first >> 3) + (size_t)lane;
        if (byte < ((nprops + 7) >> 3)) qbitmap[byte] = (uint8_t)(m >> (8 * lane));
    }
}

// Tuples per upload piece of a registered-key shard (SBV_SHARD_PIECE_KEYED).  No table is built per batch, so a piece only has to
// fill the device: 2^17 signatures = two wavefronts on every SIMD, above the 8-lanes-per-signature latency form's range.
size_t g_shard_piece_keyed = (size_t)1 << 17;

// One device's share of a sharded registered-key call; c.mu and g_reg_mu (shared) held.  The same two upload slots as verify_shard:
// 96-byte records r | s | hash and their 4-byte slots — 100 B per signature over PCIe instead of 160 — of piece i + 1 travel on
// the copy stream beside stage A + B of piece i; quorum bits by distinct slot.
// pieces of a registered-key shard: as many as the configured piece size asks for, then EQUAL sizes (a multiple of the granule) —
// a short last piece would fall into the 8-lanes-per-signature latency kernel and every piece fills the device for about one round
size_t keyed_piece(size_t m, size_t gran) {
    size_t chunk = (g_shard_piece_keyed < kMaxChunk / 2 ? g_shard_piece_keyed : kMaxChunk / 2) / gran * gran;
    if (chunk == 0) chunk = gran;
    const size_t pieces = (m + chunk - 1) / chunk;
    size_t per = ((m + pieces - 1) / pieces + gran - 1) / gran * gran;
    return per < chunk ? per : chunk;
}

int verify_shard_keyed(Context& c, const uint8_t* h_rsh, const u32* h_slots, size_t m, size_t group, u32 quorum, uint8_t* d_slot, uint8_t* d_qslot,
                       double* h2d_us, double* kern_us) {
    if (!c.ready) { g_err = "device not initialised"; return SBV_ENOTINIT; }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    int rc = sync_registry(c);                 // this device's replica of the key registry (a no-op when it is current)
    if (rc != SBV_OK) return rc;
    if (c.nkeys == 0) { g_err = "no keys registered"; return SBV_EINVAL; }
    const size_t gran = shard_granule(group);
    if (gran > kMaxChunk / 2) { g_err = "group too large"; return SBV_EINVAL; }
    const size_t chunk = keyed_piece(m, gran);
    const size_t pieces = (m + chunk - 1) / chunk;
    const bool two = pieces > 1;
    // Two pieces are in flight: piece i + 1 is uploaded (copy stream) and runs its stage A (second kernel stream, its own half of
    // the scratch planes) while stage B of piece i still fills the device — stage A is a short latency-bound chain that would
    // otherwise stand between two stage-B launches on one stream (measured: 5 pieces of 2^17 on one stream 1.57 ms of kernels for
    // 550 000 signatures against 1.05 ms in one launch; profiles/r05/).
    rc = ensure_capacity(c, two ? 2 * chunk : m);
    if (rc != SBV_OK) return rc;
    ShardBuffers& sbuf = g_shard[c.device];
    if (two) {
        rc = grow_bytes(sbuf.d_stage2, sbuf.stage2_cap, chunk * SBV_TUPLE_BYTES);
        if (rc == SBV_OK) rc = grow_bytes(sbuf.d_slots2, sbuf.slots2_cap, chunk * sizeof(u32));
        if (rc != SBV_OK) return rc;
        if (!c.copy_stream && hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking) != hipSuccess) { g_err = "hipStreamCreate failed"; return SBV_EDEVICE; }
    }
    hipStream_t up = two ? c.copy_stream : c.stream;
    hipStream_t ks[2] = {c.stream, two ? c.gsync.side_a : c.stream};      // the grouped step's side stream is idle on this path (a process gets four hardware queues)
    uint8_t* stage[2] = {c.d_tuples, two ? sbuf.d_stage2 : c.d_tuples};
    u32* sstage[2] = {c.d_slots, two ? reinterpret_cast<u32*>(sbuf.d_slots2) : c.d_slots};
    std::vector<hipEvent_t> ev(3 * pieces, nullptr);
    auto drop_events = [&] { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); };
    for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) { drop_events(); g_err = "hipEventCreate failed"; return SBV_EDEVICE; }
    hipError_t he = hipSuccess;
    auto step = [&](hipError_t r) { if (he == hipSuccess) he = r; };
    if (c.busy_valid) {
        step(hipStreamWaitEvent(ks[0], c.busy, 0));
        if (two) { step(hipStreamWaitEvent(ks[1], c.busy, 0)); step(hipStreamWaitEvent(up, c.busy, 0)); }
    }
    size_t i = 0;
    for (size_t off = 0; off < m && rc == SBV_OK && he == hipSuccess; off += chunk, ++i) {
        const size_t k = m - off < chunk ? m - off : chunk;
        const int w = (int)(i & 1);
        if (i >= 2) step(hipStreamWaitEvent(up, ev[3 * (i - 2) + 2], 0));       // the staging set's previous kernels are done with it
        step(hipEventRecord(ev[3 * i], up));
        step(hipMemcpyAsync(stage[w], h_rsh + off * 96, k * 96, hipMemcpyHostToDevice, up));
        step(hipMemcpyAsync(sstage[w], h_slots + off, k * sizeof(u32), hipMemcpyHostToDevice, up));
        step(hipEventRecord(ev[3 * i + 1], up));
        if (two) step(hipStreamWaitEvent(ks[w], ev[3 * i + 1], 0));
        if (he != hipSuccess) break;
        rc = enqueue_keyed(c, stage[w], sstage[w], k, d_slot + off / 8, ks[w], nullptr, w ? chunk : 0);
        if (rc != SBV_OK) break;
        if (d_qslot && group > 0 && quorum > 0) {
            const size_t props = k / group;
            if (props)
                hipLaunchKernelGGL(k_quorum_bits_slots, dim3((unsigned)((props + 255) / 256)), dim3(256), 0, ks[w], sstage[w], d_slot + off / 8,
                                   props, (u32)group, quorum, d_qslot + (off / group) / 8);
        }
        step(hipEventRecord(ev[3 * i + 2], ks[w]));
    }
    // drain in every case: the staging sets and the caller's host buffers must not be in use when this returns
    if (two) { step(hipStreamSynchronize(up)); step(hipStreamSynchronize(ks[1])); }
    step(hipStreamSynchronize(ks[0]));
    if (rc == SBV_OK && he != hipSuccess) rc = fail(SBV_EDEVICE, "verify_shard_keyed", he);
    if (rc == SBV_OK) {
        if (h2d_us) for (size_t j = 0; j < pieces; ++j) *h2d_us += 1e3 * ms_between(ev[3 * j], ev[3 * j + 1]);
        if (kern_us) *kern_us += 1e3 * ms_between(ev[1], ev[3 * (pieces - 1) + 2]);
    }
    drop_events();
    c.busy_valid = false;
    return rc;
}

// One device's share of a sharded raw-messages call (SURVEY.md section 8f row 1 across the node); c.mu and g_reg_mu (shared) held.
// Piece i = signatures [first + i * chunk, ...): its message bytes, DER signatures, the slices of the caller's two offset tables as
// they are (the kernel subtracts the slice's base) and its key slots go up on the copy stream into one of two staging sets while
// SHA-256 + DER (k_msg_frontend), stage A and stage B of the previous piece run on c.stream.  The front end's records land in
// Context::d_tuples: kernels of all pieces are ordered on one stream, so one record buffer serves.
int verify_shard_msgs(Context& c, const uint8_t* msgs, const uint64_t* moff, const uint8_t* sigs, const uint64_t* soff, const u32* h_slots,
                      size_t first, size_t m, size_t group, u32 quorum, uint8_t* d_slot, uint8_t* d_qslot, double* h2d_us, double* kern_us) {
    if (!c.ready) { g_err = "device not initialised"; return SBV_ENOTINIT; }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    int rc = sync_registry(c);
    if (rc != SBV_OK) return rc;
    if (c.nkeys == 0) { g_err = "no keys registered"; return SBV_EINVAL; }
    const size_t gran = shard_granule(group);
    if (gran > kMaxChunk / 2) { g_err = "group too large"; return SBV_EINVAL; }
    const size_t chunk = keyed_piece(m, gran);
    const size_t pieces = (m + chunk - 1) / chunk;
    size_t max_mb = 0, max_sb = 0;                 // staging for the largest piece, grown before anything is enqueued
    for (size_t off = 0; off < m; off += chunk) {
        const size_t k = m - off < chunk ? m - off : chunk;
        // the piece boundaries of the caller's tables size the uploads: they must be monotone and plausible (a piece of k signatures
        // with more than 64 KiB per message / 4 KiB per signature on average is a corrupt table, not a batch); everything between two
        // boundaries is checked lane by lane on the device (k_msg_frontend)
        if (moff[first + off + k] < moff[first + off] || soff[first + off + k] < soff[first + off] ||
            moff[first + off + k] - moff[first + off] > (uint64_t)k * 65536 || soff[first + off + k] - soff[first + off] > (uint64_t)k * 4096) {
            g_err = "offset table is not monotone";
            return SBV_EINVAL;
        }
        const size_t mb = (size_t)(moff[first + off + k] - moff[first + off]), sb = (size_t)(soff[first + off + k] - soff[first + off]);
        if (mb > max_mb) max_mb = mb;
        if (sb > max_sb) max_sb = sb;
    }
    const size_t kmax = m < chunk ? m : chunk;
    rc = ensure_capacity(c, pieces > 1 ? 2 * chunk : kmax);        // two pieces in flight, each in its half of the scratch planes and of the record buffer
    if (rc == SBV_OK) rc = grow(c.d_msgs, c.msgs_cap, max_mb + 16);
    if (rc == SBV_OK) rc = grow(c.d_sigs, c.sigs_cap, max_sb + 16);
    if (rc == SBV_OK) rc = grow(c.d_moff, c.moff_cap, kmax + 1);
    if (rc == SBV_OK) rc = grow(c.d_soff, c.soff_cap, kmax + 1);
    ShardBuffers& sbuf = g_shard[c.device];
    if (rc == SBV_OK && pieces > 1) {
        rc = grow_bytes(sbuf.d_msgs2, sbuf.msgs2_cap, max_mb + 16);
        if (rc == SBV_OK) rc = grow_bytes(sbuf.d_sigs2, sbuf.sigs2_cap, max_sb + 16);
        if (rc == SBV_OK) rc = grow_bytes(sbuf.d_moff2, sbuf.moff2_cap, (kmax + 1) * sizeof(uint64_t));
        if (rc == SBV_OK) rc = grow_bytes(sbuf.d_soff2, sbuf.soff2_cap, (kmax + 1) * sizeof(uint64_t));
        if (rc == SBV_OK) rc = grow_bytes(sbuf.d_slots2, sbuf.slots2_cap, kmax * sizeof(u32));
        if (rc == SBV_OK && !c.copy_stream && hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking) != hipSuccess) { g_err = "hipStreamCreate failed"; rc = SBV_EDEVICE; }
    }
    if (rc != SBV_OK) return rc;
    hipStream_t up = pieces > 1 ? c.copy_stream : c.stream;
    const bool two = pieces > 1;
    hipStream_t ks[2] = {c.stream, two ? c.gsync.side_a : c.stream};      // as verify_shard_keyed: front end + stage A of piece i + 1 beside stage B of piece i
    uint8_t* recs[2] = {c.d_tuples, two ? c.d_tuples + chunk * 96 : c.d_tuples};      // the front end's 96-byte records (Context::d_tuples holds 160 bytes per scratch slot)
    uint8_t* st_msgs[2] = {c.d_msgs, two ? sbuf.d_msgs2 : c.d_msgs};
    uint8_t* st_sigs[2] = {c.d_sigs, two ? sbuf.d_sigs2 : c.d_sigs};
    uint64_t* st_moff[2] = {c.d_moff, two ? reinterpret_cast<uint64_t*>(sbuf.d_moff2) : c.d_moff};
    uint64_t* st_soff[2] = {c.d_soff, two ? reinterpret_cast<uint64_t*>(sbuf.d_soff2) : c.d_soff};
    u32* st_slots[2] = {c.d_slots, two ? reinterpret_cast<u32*>(sbuf.d_slots2) : c.d_slots};
    std::vector<hipEvent_t> ev(3 * pieces, nullptr);
    auto drop_events = [&] { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); };
    for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) { drop_events(); g_err = "hipEventCreate failed"; return SBV_EDEVICE; }
    hipError_t he = hipSuccess;
    auto step = [&](hipError_t r) { if (he == hipSuccess) he = r; };
    if (c.busy_valid) {
        step(hipStreamWaitEvent(ks[0], c.busy, 0));
        if (two) { step(hipStreamWaitEvent(ks[1], c.busy, 0)); step(hipStreamWaitEvent(up, c.busy, 0)); }
    }
    size_t i = 0;
    for (size_t off = 0; off < m && rc == SBV_OK && he == hipSuccess; off += chunk, ++i) {
        const size_t k = m - off < chunk ? m - off : chunk;
        const size_t a = first + off;
        const uint64_t mbase = moff[a], sbase = soff[a];
        const size_t mb = (size_t)(moff[a + k] - mbase), sb = (size_t)(soff[a + k] - sbase);
        const int w = (int)(i & 1);
        if (i >= 2) step(hipStreamWaitEvent(up, ev[3 * (i - 2) + 2], 0));       // the staging set's previous kernels are done with it
        step(hipEventRecord(ev[3 * i], up));
        if (mb) step(hipMemcpyAsync(st_msgs[w], msgs + mbase, mb, hipMemcpyHostToDevice, up));
        if (sb) step(hipMemcpyAsync(st_sigs[w], sigs + sbase, sb, hipMemcpyHostToDevice, up));
        step(hipMemcpyAsync(st_moff[w], moff + a, (k + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, up));
        step(hipMemcpyAsync(st_soff[w], soff + a, (k + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, up));
        step(hipMemcpyAsync(st_slots[w], h_slots + a, k * sizeof(u32), hipMemcpyHostToDevice, up));
        step(hipEventRecord(ev[3 * i + 1], up));
        if (two) step(hipStreamWaitEvent(ks[w], ev[3 * i + 1], 0));
        if (he != hipSuccess) break;
        step(sbv::launch_msg_frontend(st_msgs[w], st_moff[w], st_sigs[w], st_soff[w], k, reinterpret_cast<u32*>(recs[w]), ks[w], mbase, sbase, mb, sb));
        if (he != hipSuccess) break;
        rc = enqueue_keyed(c, recs[w], st_slots[w], k, d_slot + off / 8, ks[w], nullptr, w ? chunk : 0);
        if (rc != SBV_OK) break;
        if (d_qslot && group > 0 && quorum > 0) {
            const size_t props = k / group;
            if (props)
                hipLaunchKernelGGL(k_quorum_bits_slots, dim3((unsigned)((props + 255) / 256)), dim3(256), 0, ks[w], st_slots[w], d_slot + off / 8,
                                   props, (u32)group, quorum, d_qslot + (off / group) / 8);
        }
        step(hipEventRecord(ev[3 * i + 2], ks[w]));
    }
    if (two) { step(hipStreamSynchronize(up)); step(hipStreamSynchronize(ks[1])); }
    step(hipStreamSynchronize(ks[0]));
    if (rc == SBV_OK && he != hipSuccess) rc = fail(SBV_EDEVICE, "verify_shard_msgs", he);
    if (rc == SBV_OK) {
        if (h2d_us) for (size_t j = 0; j < pieces; ++j) *h2d_us += 1e3 * ms_between(ev[3 * j], ev[3 * j + 1]);
        if (kern_us) *kern_us += 1e3 * ms_between(ev[1], ev[3 * (pieces - 1) + 2]);
    }
    drop_events();
    c.busy_valid = false;
    return rc;
}

}  // namespace

// The per-device minimum the sharded entry uses for THIS batch (host memory, no device): the keys of 256 evenly spaced tuples
// are compared; at most 32 distinct ones among them = a consenter-style batch (a batch over K equally likely signers shows
// about min(K, 256 (1 - e^(-256/K)) ...) distinct keys in such a sample: 16 for K = 16, 226 for K = 1024).
extern "C" size_t sbv_shard_min_for(const uint8_t* tuples, size_t n, size_t group) {
    (void)group;
    if (g_shard_min_env || !tuples || n == 0) return g_shard_min;
    const size_t samples = n < 256 ? n : 256;
    const uint8_t* seen[33];
    size_t distinct = 0;
    for (size_t j = 0; j < samples && distinct <= 32; ++j) {
        const uint8_t* k = tuples + (j * n / samples) * SBV_TUPLE_BYTES + 96;
        bool dup = false;
        for (size_t d = 0; d < distinct && !dup; ++d) dup = memcmp(seen[d], k, 64) == 0;
        if (!dup) seen[distinct++] = k;
    }
    return distinct <= 32 ? g_shard_min_few : g_shard_min;
}

extern "C" size_t sbv_shard_plan(size_t n, int devices, size_t group, size_t min_per_device, size_t* first) {
    // first[0..shards] = tuple index where each shard starts (first[shards] = n); returns the number of shards (>= 1)
    if (devices < 1) devices = 1;
    if (devices > kMaxDevices) devices = kMaxDevices;
    if (min_per_device == 0) min_per_device = g_shard_min;
    const size_t gran = shard_granule(group);
    size_t shards = 1;
    if (devices > 1 && n >= 2 * min_per_device) {
        shards = n / min_per_device;
        if (shards > (size_t)devices) shards = (size_t)devices;
    }
    size_t per = ((n + shards - 1) / shards + gran - 1) / gran * gran;
    if (per == 0) per = gran;
    shards = (n + per - 1) / per;
    if (shards == 0) shards = 1;
    if (first) {
        for (size_t k = 0; k <= shards; ++k) { const size_t f = k * per; first[k] = f < n ? f : n; }
    }
    return shards;
}

extern "C" int sbv_init_all(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_err = "no HIP device visible (libsbv has no CPU fallback)";
        return SBV_ENODEV;
    }
    ndev = logical_devices(ndev);
    // Every device from its own host thread (round 6; VERDICT r5 #1c): a context's start-up is two table uploads (0.47 GB) and a dozen
    // allocations, independent per device — 8 GPUs came up one after another before.  The host tables are built once (call_once).
    std::vector<int> rcs((size_t)ndev, SBV_OK);
    std::vector<std::string> errs((size_t)ndev);
    {
        std::vector<std::thread> th;
        for (int d = 0; d < ndev; ++d)
            th.emplace_back([&, d] { rcs[(size_t)d] = sbv_init(d); if (rcs[(size_t)d] != SBV_OK) errs[(size_t)d] = g_err; });
        for (auto& t : th) t.join();
    }
    std::vector<int> devs;
    for (int d = 0; d < ndev; ++d) if (rcs[(size_t)d] == SBV_OK) devs.push_back(d);
    if (devs.empty()) { g_err = errs.back(); return rcs.back(); }       // not a single usable device
    std::lock_guard<std::mutex> lk(g_mu);
    {   // sbv_init made the first context to FINISH the default; with every device up, the default is the lowest index again
        if (g_ctxs[devs[0]]) g_def = g_ctxs[devs[0]].get();
    }
    if (g_devs != devs) {
        rccl_teardown();
        g_devs = devs;
    }
    if (const char* e = getenv("SBV_SHARD_MIN")) { const long v = atol(e); if (v > 0) { g_shard_min = g_shard_min_few = (size_t)v; g_shard_min_env = true; } }
    if (const char* e = getenv("SBV_SHARD_PIECE")) { const long v = atol(e); if (v >= 512) g_shard_piece = (size_t)v; }
    if (const char* e = getenv("SBV_SHARD_PIECE_KEYED")) { const long v = atol(e); if (v >= 512) g_shard_piece_keyed = (size_t)v; }
    if (const char* e = getenv("SBV_SHARD_MODE")) g_shard_mode.store(strcmp(e, "keys") == 0 ? 1 : 0);
    if (const char* e = getenv("SBV_SHARD_PARTS")) { const long v = atol(e); if (v >= 0 && v <= 64) g_shard_parts.store((unsigned)v); }
    const char* force = getenv("SBV_RCCL");
    if (g_devs.size() > 1 || (force && force[0] == '1')) (void)rccl_setup();     // failure -> host-side gather
    return (int)g_devs.size();
}

extern "C" int sbv_initialised_devices(int* out, int max) {
    std::lock_guard<std::mutex> lk(g_mu);
    int k = 0;
    for (int d = 0; d < kMaxDevices; ++d) {
        if (!g_ctxs[d] || !g_ctxs[d]->ready) continue;
        if (out && k < max) out[k] = d;
        ++k;
    }
    return k;
}

extern "C" int sbv_shard_mode(int by_key, unsigned parts) {
    g_shard_mode.store(by_key ? 1 : 0);
    g_shard_parts.store(parts > 64 ? 64u : parts);
    return SBV_OK;
}

namespace {
// The key-affine form of the sharded entry: every participating device receives the whole batch over its own PCIe link,
// verifies the tuples of its parts (part p runs on device p % devices) and leaves their verdict bits in a full-size bitmap
// of its own; the bitmaps have disjoint bits, so their OR is their byte-wise SUM: one in-place ncclAllReduce(sum, uint8)
// when RCCL is up, an OR on the host otherwise.  The per-proposal quorum bits need all of a proposal's verdicts, which now
// live on different devices: they are computed on the first device from the combined bitmap.
int sharded_by_key(const uint8_t* tuples, size_t n, size_t group, u32 quorum, uint8_t* accept_bitmap, uint8_t* quorum_bitmap,
                   sbv_shard_info* info, const std::vector<int>& devs, unsigned parts, bool use_rccl) {
    const auto t0 = std::chrono::steady_clock::now();
    const size_t ndev = devs.size() < parts ? devs.size() : parts;
    const size_t words = (n + 31) / 32, bytes = (n + 7) / 8;
    std::vector<int> rcs(ndev, SBV_OK);
    std::vector<std::string> errs(ndev);
    std::vector<double> h2d(ndev, 0.0), kern(ndev, 0.0);
    std::vector<std::vector<uint8_t>> host_bits(use_rccl ? 0 : ndev);
    std::unique_lock<std::mutex> sharded_lk(g_sharded_mu);
    auto work = [&](size_t d) {
        Context* c;
        { std::lock_guard<std::mutex> lk(g_mu); c = g_ctxs[devs[d]].get(); }
        std::lock_guard<std::mutex> lkc(c->mu);
        ShardBuffers& sbuf = g_shard[devs[d]];
        int rc = SBV_OK;
        auto step = [&](hipError_t e, const char* what) { if (rc == SBV_OK && e != hipSuccess) rc = fail(SBV_EDEVICE, what, e); };
        if (!c->ready) { g_err = "device not initialised"; rc = SBV_ENOTINIT; }
        if (rc == SBV_OK) step(hipSetDevice(c->hip_dev), "hipSetDevice");
        if (rc == SBV_OK) rc = grow_bytes(sbuf.d_full, sbuf.full_cap, n * SBV_TUPLE_BYTES + 64);
        if (rc == SBV_OK) rc = grow_bytes(sbuf.d_gather, sbuf.gather_cap, words * 4 + 64);
        if (rc == SBV_OK) {
            step(hipEventRecord(c->ev[0], c->stream), "hipEventRecord");
            step(hipMemcpyAsync(sbuf.d_full, tuples, n * SBV_TUPLE_BYTES, hipMemcpyHostToDevice, c->stream), "H2D of the batch");
            step(hipEventRecord(c->ev[1], c->stream), "hipEventRecord");
            step(hipMemsetAsync(sbuf.d_gather, 0, words * 4, c->stream), "hipMemsetAsync");
        }
        for (unsigned p = (unsigned)d; rc == SBV_OK && p < parts; p += (unsigned)ndev)
            rc = part_enqueue(*c, sbuf.d_full, n, p, parts, reinterpret_cast<u32*>(sbuf.d_gather), c->stream, nullptr, false);
        if (rc == SBV_OK) {
            step(hipEventRecord(c->ev[3], c->stream), "hipEventRecord");
            if (!use_rccl) {
                host_bits[d].resize(bytes);
                step(hipMemcpyAsync(host_bits[d].data(), sbuf.d_gather, bytes, hipMemcpyDeviceToHost, c->stream), "D2H of a device's bitmap");
            }
            step(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
            if (rc == SBV_OK) { h2d[d] = 1e3 * ms_between(c->ev[0], c->ev[1]); kern[d] = 1e3 * ms_between(c->ev[1], c->ev[3]); }
        }
        rcs[d] = rc;
        if (rc != SBV_OK) errs[d] = g_err;
    };
    if (ndev == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (size_t d = 0; d < ndev; ++d) th.emplace_back(work, d);
        for (auto& t : th) t.join();
    }
    for (size_t d = 0; d < ndev; ++d)
        if (rcs[d] != SBV_OK) { g_err = "device " + std::to_string(devs[d]) + ": " + errs[d]; return rcs[d]; }
    const auto t1 = std::chrono::steady_clock::now();
    Context* c0;
    { std::lock_guard<std::mutex> lk(g_mu); c0 = g_ctxs[devs[0]].get(); }
    if (use_rccl) {
        std::lock_guard<std::mutex> lk(g_mu);                    // the communicators
        // Every rank of the communicator takes part (devs == g_devs: checked by the caller); with fewer parts than devices the
        // idle ranks contribute a zeroed bitmap (this used to fail outright with parts < devices).
        const size_t world = devs.size();
        bool ok = g_rccl.ready && g_rccl.AllReduce && g_rccl.comms.size() == world;
        for (size_t d = ndev; ok && d < world; ++d) {
            ShardBuffers& sbuf = g_shard[devs[d]];
            ok = hipSetDevice(hipdev_of(devs[d])) == hipSuccess && grow_bytes(sbuf.d_gather, sbuf.gather_cap, words * 4 + 64) == SBV_OK &&
                 hipMemsetAsync(sbuf.d_gather, 0, words * 4, g_ctxs[devs[d]]->stream) == hipSuccess;
        }
        ok = ok && g_rccl.GroupStart() == 0;
        for (size_t d = 0; ok && d < world; ++d) {
            ok = hipSetDevice(hipdev_of(devs[d])) == hipSuccess;
            ShardBuffers& sbuf = g_shard[devs[d]];
            if (ok) ok = g_rccl.AllReduce(sbuf.d_gather, sbuf.d_gather, words * 4, /*ncclUint8*/ 1, /*ncclSum*/ 0, g_rccl.comms[d], g_ctxs[devs[d]]->stream) == 0;
        }
        if (ok) ok = g_rccl.GroupEnd() == 0;
        if (ok) ok = hipSetDevice(hipdev_of(devs[0])) == hipSuccess &&
                     hipMemcpyAsync(accept_bitmap, g_shard[devs[0]].d_gather, bytes, hipMemcpyDeviceToHost, c0->stream) == hipSuccess &&
                     hipStreamSynchronize(c0->stream) == hipSuccess;
        for (size_t d = 1; ok && d < world; ++d) ok = hipSetDevice(hipdev_of(devs[d])) == hipSuccess && hipStreamSynchronize(g_ctxs[devs[d]]->stream) == hipSuccess;
        if (!ok) { g_err = "RCCL all-reduce of the per-device bitmaps failed"; return SBV_EDEVICE; }
    } else {
        memcpy(accept_bitmap, host_bits[0].data(), bytes);
        for (size_t d = 1; d < ndev; ++d)
            for (size_t b = 0; b < bytes; ++b) accept_bitmap[b] |= host_bits[d][b];
    }
    if (quorum_bitmap && group > 0 && quorum > 0 && n / group > 0) {
        std::lock_guard<std::mutex> lkc(c0->mu);
        ShardBuffers& sbuf = g_shard[devs[0]];
        const size_t props = n / group;
        HIP_TRY(SBV_EDEVICE, hipSetDevice(c0->hip_dev));
        int rc = grow_bytes(sbuf.d_q, sbuf.q_cap, (props + 7) / 8 + 64);
        if (rc != SBV_OK) return rc;
        if (!use_rccl) HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(sbuf.d_gather, accept_bitmap, bytes, hipMemcpyHostToDevice, c0->stream));
        hipLaunchKernelGGL(k_quorum_bits, dim3((unsigned)((props + 255) / 256)), dim3(256), 0, c0->stream, sbuf.d_full, sbuf.d_gather, props, (u32)group, quorum, sbuf.d_q);
        HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(quorum_bitmap, sbuf.d_q, (props + 7) / 8, hipMemcpyDeviceToHost, c0->stream));
        HIP_TRY(SBV_EDEVICE, hipStreamSynchronize(c0->stream));
    }
    if (info) {
        info->devices = (int)devs.size();
        info->shards = (int)parts;
        info->mode = use_rccl ? 3 : 4;
        info->tuples_per_shard = (n + parts - 1) / parts;
        for (size_t d = 0; d < ndev; ++d) { if (h2d[d] > info->h2d_us) info->h2d_us = h2d[d]; if (kern[d] > info->kernels_us) info->kernels_us = kern[d]; }
        info->gather_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
        info->total_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    return SBV_OK;
}
}  // namespace

namespace {
// The contiguous partition shared by the generic and the registered-key sharded entries: plan, one host thread per shard (each
// under its device's lock, running `shard_fn` on tuples [first, first + count) with the device's slot of the gather buffer and its
// quorum slot), the in-place all-gather when more than one device took part, the final copies.
using ShardFn = std::function<int(Context&, size_t first, size_t count, uint8_t* d_bits, uint8_t* d_q, double* h2d_us, double* kern_us)>;
int sharded_contiguous(size_t n, size_t group, u32 quorum, uint8_t* accept_bitmap, uint8_t* quorum_bitmap, sbv_shard_info* info,
                       const std::vector<int>& devs, bool use_rccl, size_t min_per_device, const ShardFn& shard_fn,
                       std::chrono::steady_clock::time_point t0) {
    size_t first[kMaxDevices + 1];
    const size_t shards = sbv_shard_plan(n, (int)devs.size(), group, min_per_device, first);
    const size_t per = shards > 1 ? first[1] - first[0] : ((n + shard_granule(group) - 1) / shard_granule(group) * shard_granule(group));
    const size_t sb = per / 8;                                          // bitmap bytes per shard slot
    const size_t qb = group ? (per / group + 7) / 8 : 0;                // quorum bytes per shard slot
    // which device takes which shard: all of them in order, or one picked round-robin for an unsplit batch
    std::vector<int> use(shards);
    if (shards == 1) use[0] = devs[g_rr.fetch_add(1) % devs.size()];
    else for (size_t k = 0; k < shards; ++k) use[k] = devs[k];
    // The collective runs whenever more than one device took part.  The communicator spans every initialised device, so with
    // fewer shards than devices the idle ranks join the all-gather with an unused slot (one tiny launch on an idle GPU; no
    // sub-communicators to create and cache at run time): slots [0, shards) of the gathered buffer are the bitmap.
    const bool gather = shards > 1 && use_rccl;
    const size_t ranks = gather ? devs.size() : shards;
    const bool forced_single = shards == 1 && use_rccl && devs.size() == 1;   // SBV_RCCL=1 on a one-GPU box: exercise the collective with one rank
    std::vector<int> rcs(shards, SBV_OK);
    std::vector<std::string> errs(shards);
    std::vector<double> h2d(shards, 0.0), kern(shards, 0.0);
    // A call that spans devices owns every device's gather buffer until its final copy; a call that stays on one device
    // (replica route) uses that device's private buffer under its context lock and runs beside other replicas.
    const bool multi = shards > 1 || forced_single;
    std::unique_lock<std::mutex> sharded_lk(g_sharded_mu, std::defer_lock);
    if (multi) sharded_lk.lock();
    auto work = [&](size_t k) {
        Context* c;
        { std::lock_guard<std::mutex> lk(g_mu); c = g_ctxs[use[k]].get(); }
        std::lock_guard<std::mutex> lkc(c->mu);
        ShardBuffers& sbuf = g_shard[use[k]];
        uint8_t*& d_bits = multi ? sbuf.d_gather : sbuf.d_on;
        size_t& bits_cap = multi ? sbuf.gather_cap : sbuf.on_cap;
        uint8_t*& d_qb = multi ? sbuf.d_q : sbuf.d_onq;
        size_t& qb_cap = multi ? sbuf.q_cap : sbuf.onq_cap;
        int rc = SBV_OK;
        if (hipSetDevice(c->hip_dev) != hipSuccess) rc = SBV_EDEVICE;
        if (rc == SBV_OK) rc = grow_bytes(d_bits, bits_cap, sb * ranks + 64);
        if (rc == SBV_OK && quorum_bitmap) rc = grow_bytes(d_qb, qb_cap, qb + 64);
        if (rc == SBV_OK)
            rc = shard_fn(*c, first[k], first[k + 1] - first[k], d_bits + k * sb, quorum_bitmap ? d_qb : nullptr, &h2d[k], &kern[k]);
        if (rc == SBV_OK && quorum_bitmap) {
            const size_t props = (first[k + 1] - first[k]) / group;
            if (props && hipMemcpy(quorum_bitmap + (first[k] / group) / 8, d_qb, (props + 7) / 8, hipMemcpyDeviceToHost) != hipSuccess) rc = SBV_EDEVICE;
        }
        if (rc == SBV_OK && !gather && !forced_single) {         // no collective: this shard's bitmap goes straight to the host
            const size_t bytes = (first[k + 1] - first[k] + 7) / 8;
            if (hipMemcpy(accept_bitmap + first[k] / 8, d_bits + k * sb, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = SBV_EDEVICE;
        }
        rcs[k] = rc;
        if (rc != SBV_OK) errs[k] = g_err;
    };
    if (shards == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (size_t k = 0; k < shards; ++k) th.emplace_back(work, k);
        for (auto& t : th) t.join();
    }
    for (size_t k = 0; k < shards; ++k)
        if (rcs[k] != SBV_OK) { g_err = "device " + std::to_string(use[k]) + ": " + errs[k]; return rcs[k]; }
    const auto t1 = std::chrono::steady_clock::now();
    double gather_us = 0;
    if (gather || forced_single) {
        std::lock_guard<std::mutex> lk(g_mu);                    // the communicators
        // rank r of the communicator is device g_devs[r]; shard k ran on devs[k] = g_devs[k], so slot k IS rank k's slot (the
        // in-place form needs sendbuff = recvbuff + rank * count).  Ranks [shards, world) are idle devices: their slot is unused.
        const size_t world = forced_single ? 1 : g_devs.size();
        bool ok = g_rccl.ready && g_rccl.comms.size() >= world;
        for (size_t r = shards; ok && r < world; ++r)                  // an idle rank needs a buffer of its own for the gathered slots
            ok = hipSetDevice(hipdev_of(g_devs[r])) == hipSuccess && grow_bytes(g_shard[g_devs[r]].d_gather, g_shard[g_devs[r]].gather_cap, sb * world + 64) == SBV_OK;
        ok = ok && g_rccl.GroupStart() == 0;
        for (size_t r = 0; ok && r < world; ++r) {
            const int dev = forced_single ? use[0] : g_devs[r];
            ok = (r >= shards || use[r] == dev) && hipSetDevice(hipdev_of(dev)) == hipSuccess;
            ShardBuffers& sbuf = g_shard[dev];
            if (ok) ok = g_rccl.AllGather(sbuf.d_gather + r * sb, sbuf.d_gather, sb, /*ncclUint8*/ 1, g_rccl.comms[r], g_ctxs[dev]->stream) == 0;
        }
        if (ok) ok = g_rccl.GroupEnd() == 0;
        if (ok) ok = hipSetDevice(hipdev_of(use[0])) == hipSuccess &&
                     hipMemcpyAsync(accept_bitmap, g_shard[use[0]].d_gather, (n + 7) / 8, hipMemcpyDeviceToHost, g_ctxs[use[0]]->stream) == hipSuccess &&
                     hipStreamSynchronize(g_ctxs[use[0]]->stream) == hipSuccess;
        for (size_t r = 1; ok && r < world; ++r) ok = hipSetDevice(hipdev_of(g_devs[r])) == hipSuccess && hipStreamSynchronize(g_ctxs[g_devs[r]]->stream) == hipSuccess;
        if (!ok) { g_err = "RCCL all-gather of the bitmap shards failed"; return SBV_EDEVICE; }
        gather_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
    }
    if (info) {
        info->devices = (int)devs.size();
        info->shards = (int)shards;
        info->mode = (gather || forced_single) ? 1 : (shards > 1 ? 2 : 0);
        info->tuples_per_shard = per;
        for (size_t k = 0; k < shards; ++k) { if (h2d[k] > info->h2d_us) info->h2d_us = h2d[k]; if (kern[k] > info->kernels_us) info->kernels_us = kern[k]; }
        info->gather_us = gather_us;
        info->total_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    return SBV_OK;
}
}  // namespace

extern "C" int sbv_p256_verify_batch_sharded(const uint8_t* tuples, size_t n, size_t group, uint32_t quorum, uint8_t* accept_bitmap,
                                             uint8_t* quorum_bitmap, sbv_shard_info* info) {
    if (info) memset(info, 0, sizeof *info);
    if (n == 0) return SBV_OK;
    if (!tuples || !accept_bitmap) { g_err = "null pointer"; return SBV_EINVAL; }
    if (quorum_bitmap && (group == 0 || quorum == 0 || quorum > group || group > 64)) { g_err = "quorum bits need 0 < quorum <= group <= 64"; return SBV_EINVAL; }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<int> devs;
    bool use_rccl;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        devs = g_devs;
        if (devs.empty() && g_def) devs.push_back(g_def->device);     // sbv_init only: one device
        use_rccl = g_rccl.ready && g_rccl.comms.size() == devs.size();   // the communicator spans exactly these devices
    }
    if (devs.empty()) { g_err = "sbv_init_all / sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (g_shard_mode.load() == 1) {
        unsigned parts = g_shard_parts.load();
        if (parts == 0) parts = (unsigned)devs.size();
        if (parts > 1 && n >= parts) return sharded_by_key(tuples, n, group, quorum, accept_bitmap, quorum_bitmap, info, devs, parts, use_rccl);
    }
    return sharded_contiguous(n, group, quorum, accept_bitmap, quorum_bitmap, info, devs, use_rccl, sbv_shard_min_for(tuples, n, group),
                              [&](Context& c, size_t first, size_t count, uint8_t* d_bits, uint8_t* d_q, double* h2d_us, double* kern_us) {
                                  return verify_shard(c, tuples + first * SBV_TUPLE_BYTES, count, group, quorum, d_bits, d_q, h2d_us, kern_us);
                              }, t0);
}

// The registered-key form of the sharded entry (round 5; include/sbv.h): the same plan, pieces, quorum bits and collective over
// 96-byte records + key slots, every device holding a replica of the registry and of the consenters' wide combs.
extern "C" int sbv_p256_verify_batch_keyed_sharded(const uint8_t* rsh, const uint32_t* slots, size_t n, size_t group, uint32_t quorum,
                                                   uint8_t* accept_bitmap, uint8_t* quorum_bitmap, sbv_shard_info* info) {
    if (info) memset(info, 0, sizeof *info);
    if (n == 0) return SBV_OK;
    if (!rsh || !slots || !accept_bitmap) { g_err = "null pointer"; return SBV_EINVAL; }
    if (quorum_bitmap && (group == 0 || quorum == 0 || quorum > group || group > 64)) { g_err = "quorum bits need 0 < quorum <= group <= 64"; return SBV_EINVAL; }
    const auto t0 = std::chrono::steady_clock::now();
    std::shared_lock<std::shared_mutex> rl(g_reg_mu);        // the registry does not change under a call that replicates / reads it
    std::vector<int> devs;
    bool use_rccl;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        devs = g_devs;
        if (devs.empty() && g_def) devs.push_back(g_def->device);
        use_rccl = g_rccl.ready && g_rccl.comms.size() == devs.size();
    }
    if (devs.empty()) { g_err = "sbv_init_all / sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (g_reg.keys.empty()) { g_err = "no keys registered"; return SBV_EINVAL; }
    // nothing is built per batch on this path (the combs are resident), so the few-signers minimum applies whatever the registry holds
    return sharded_contiguous(n, group, quorum, accept_bitmap, quorum_bitmap, info, devs, use_rccl, g_shard_min_env ? g_shard_min : g_shard_min_few,
                              [&](Context& c, size_t first, size_t count, uint8_t* d_bits, uint8_t* d_q, double* h2d_us, double* kern_us) {
                                  return verify_shard_keyed(c, rsh + first * 96, slots + first, count, group, quorum, d_bits, d_q, h2d_us, kern_us);
                              }, t0);
}

// Raw messages + DER signatures + key slots, sharded (SHA-256 and the DER parse on every device's share).
extern "C" int sbv_p256_verify_msgs_keyed_sharded(const uint8_t* msgs, const uint64_t* msg_offsets, const uint8_t* sigs, const uint64_t* sig_offsets,
                                                  const uint32_t* slots, size_t n, size_t group, uint32_t quorum, uint8_t* accept_bitmap,
                                                  uint8_t* quorum_bitmap, sbv_shard_info* info) {
    if (info) memset(info, 0, sizeof *info);
    if (n == 0) return SBV_OK;
    if (!msg_offsets || !sig_offsets || !slots || !accept_bitmap) { g_err = "null pointer"; return SBV_EINVAL; }
    if (quorum_bitmap && (group == 0 || quorum == 0 || quorum > group || group > 64)) { g_err = "quorum bits need 0 < quorum <= group <= 64"; return SBV_EINVAL; }
    // The offset tables must never decrease: the piece boundaries are checked on the host (verify_shard_msgs), every entry between them by
    // its lane on the device (k_msg_frontend: an entry out of order or outside the upload becomes an empty message + signature = reject).
    // They need NOT start at 0 here: msgs / sigs are the bases the offsets refer to (a slice of a larger batch's tables is a valid argument).
    if ((msg_offsets[n] && !msgs) || (sig_offsets[n] && !sigs)) { g_err = "null pointer"; return SBV_EINVAL; }
    const auto t0 = std::chrono::steady_clock::now();
    std::shared_lock<std::shared_mutex> rl(g_reg_mu);
    std::vector<int> devs;
    bool use_rccl;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        devs = g_devs;
        if (devs.empty() && g_def) devs.push_back(g_def->device);
        use_rccl = g_rccl.ready && g_rccl.comms.size() == devs.size();
    }
    if (devs.empty()) { g_err = "sbv_init_all / sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (g_reg.keys.empty()) { g_err = "no keys registered"; return SBV_EINVAL; }
    return sharded_contiguous(n, group, quorum, accept_bitmap, quorum_bitmap, info, devs, use_rccl, g_shard_min_env ? g_shard_min : g_shard_min_few,
                              [&](Context& c, size_t first, size_t count, uint8_t* d_bits, uint8_t* d_q, double* h2d_us, double* kern_us) {
                                  return verify_shard_msgs(c, msgs, msg_offsets, sigs, sig_offsets, slots, first, count, group, quorum, d_bits, d_q, h2d_us, kern_us);
                              }, t0);
}

extern "C" int sbv_p256_verify_batch_dev_part(const void* d_tuples, size_t n, uint32_t part, uint32_t parts, void* d_bitmap_words,
                                              void* hip_stream, size_t* part_tuples) {
    SBV_ENTER(c);
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (part_tuples) *part_tuples = 0;
    if (n == 0) return SBV_OK;
    if (!d_tuples || !d_bitmap_words || (reinterpret_cast<uintptr_t>(d_tuples) & 15) || (reinterpret_cast<uintptr_t>(d_bitmap_words) & 3) ||
        parts == 0 || part >= parts) {
        g_err = "null / misaligned device pointer, or part >= parts";
        return SBV_EINVAL;
    }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    return part_enqueue(c, static_cast<const uint8_t*>(d_tuples), n, part, parts, static_cast<u32*>(d_bitmap_words), static_cast<hipStream_t>(hip_stream), part_tuples);
}

namespace {
// c.mu held: a whole host batch on this device in pieces (verify_shard), bitmap back to the host
int verify_in_pieces(Context& c, const uint8_t* tuples, size_t n, uint8_t* accept_bitmap, sbv_timing* tm) {
    ShardBuffers& sbuf = g_shard[c.device];
    if (!c.ready) { g_err = "device not initialised"; return SBV_ENOTINIT; }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.hip_dev));
    const size_t bytes = (n + 7) / 8;
    int rc = grow_bytes(sbuf.d_on, sbuf.on_cap, bytes + 64);
    if (rc != SBV_OK) return rc;
    double h2d = 0, kern = 0;
    rc = verify_shard(c, tuples, n, 0, 0, sbuf.d_on, nullptr, &h2d, &kern);
    if (rc != SBV_OK) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    HIP_TRY(SBV_EDEVICE, hipMemcpy(accept_bitmap, sbuf.d_on, bytes, hipMemcpyDeviceToHost));
    if (tm) { tm->h2d_us = h2d; tm->verify_us = kern; tm->d2h_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count(); }
    return SBV_OK;
}
}  // namespace

extern "C" int sbv_p256_verify_batch_on(int device, const uint8_t* tuples, size_t n, uint8_t* accept_bitmap) {
    if (n == 0) return SBV_OK;
    if (!tuples || !accept_bitmap) { g_err = "null pointer"; return SBV_EINVAL; }
    Context* c;
    { std::lock_guard<std::mutex> lk(g_mu); c = context_of(device, false); }
    if (!c) { g_err = "device not initialised"; return SBV_ENOTINIT; }
    std::lock_guard<std::mutex> lkc(c->mu);
    return verify_in_pieces(*c, tuples, n, accept_bitmap, nullptr);
}

extern "C" const char* sbv_last_error(void) {
    // per-thread: valid until this thread's next failing libsbv call
    return g_err.c_str();
}

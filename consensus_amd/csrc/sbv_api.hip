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

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/sbv.h"
#include "ed25519_kernels.h"
#include "p256_kernels.h"

namespace {

using sbv::u32;

struct Context {
    bool ready = false;
    int device = -1;
    size_t cap = 0;
    uint8_t* d_tuples = nullptr;
    uint8_t* d_scratch = nullptr;
    u32* d_qtab = nullptr;
    sbv::apt* d_gtab = nullptr;
    sbv::apt* d_g16r = nullptr;          // the same comb of G for the carry-free field (R = 2^261 domain)
    uint8_t* d_bitmap = nullptr;
    uint8_t* d_rerun = nullptr;         // per-wavefront flags between the fast and the exact stage-B pass
    uint8_t* h_bitmap = nullptr;        // pinned
    hipStream_t stream = nullptr;
    sbv::GroupSync gsync;               // side streams + events of the grouped stage B
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t busy = nullptr;          // completion of the last launch that used the scratch
    bool busy_valid = false;
    sbv_timing timing{};
    // in-step key grouping (p256_group.h)
    sbv::GroupBuffers grp;
    sbv::EdGroupBuffers edgrp;          // Ed25519 grouped step: per-batch combs of -A (the rest is shared with grp)
    bool group_enabled = true;
    size_t group_min_batch = 262144;    // below this the ~3.5 ms table-building latency costs more than it saves
    u32 group_min_count = 64, group_max = 2048;
    // message front end staging (grown on demand)
    uint8_t* d_msgs = nullptr; size_t msgs_cap = 0;
    uint8_t* d_sigs = nullptr; size_t sigs_cap = 0;
    uint64_t* d_moff = nullptr; uint64_t* d_soff = nullptr; size_t moff_cap = 0, soff_cap = 0;
    sbv::aniels* d_btab = nullptr;      // Ed25519 base-point comb, built on first use
    // registered keys
    sbv::apt* d_ktab = nullptr;
    uint8_t* d_kvalid = nullptr;
    u32* d_slots = nullptr;             // staging for the host-pointer keyed entry (cap entries)
    size_t key_cap = 0, nkeys = 0;
    std::unordered_map<std::string, u32> key_index;
    bool profiling = false;
    std::vector<hipEvent_t> prof_events;   // triples: before prep, after prep, after verify
    std::vector<hipEvent_t> prof_dom;      // pairs around the dominant kernel of grouped batches (nullptr pair = ungrouped)
    size_t prof_dom_used = 0;
    size_t prof_used = 0;
};

Context g_ctx;
std::mutex g_mu;
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

constexpr size_t kMaxChunk = (size_t)1 << 21;   // tuples per launch; bounds scratch at ~3.3 GB

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
    HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_qtab, want * (size_t)(SBV_QTAB_ENTRIES * 160)));
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
    std::vector<hipEvent_t*> v = {&y.ev_fork, &y.ev_assign, &y.ev_split, &y.ev_prep, &y.ev_generic};
    for (int i = 0; i < SBV_GROUP_MAX_CHUNKS; ++i) { v.push_back(&y.ev_bases[i]); v.push_back(&y.ev_tables[i]); }
    for (int i = 0; i < SBV_GROUP_MAX_SLICES; ++i) v.push_back(&y.ev_slice[i]);
    return v;
}

void free_group_buffers(Context& c) {
    sbv::GroupBuffers& b = c.grp;
    void* ptrs[] = {b.ht, b.rep, b.cnt, b.slot_of, b.group_rep, b.counters, b.grp_idx, b.ung_idx, b.slots, b.jbases, b.bases, b.jstate, b.ktab, b.kvalid, b.tmp, b.acc, b.gacc};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    b = sbv::GroupBuffers();
    if (c.edgrp.ktab) (void)hipFree(c.edgrp.ktab);
    if (c.edgrp.okb) (void)hipFree(c.edgrp.okb);
    c.edgrp = sbv::EdGroupBuffers();
}

int ensure_group_buffers(Context& c, size_t n) {
    sbv::GroupBuffers& b = c.grp;
    if (b.cap >= n && b.max_groups == c.group_max && b.gacc_cap == c.cap) { b.min_count = c.group_min_count; return SBV_OK; }
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    free_group_buffers(c);
    const size_t cap = (n + 1023) & ~(size_t)1023;
    size_t ht = 1024;
    while (ht < 2 * cap) ht *= 2;
    const size_t G = c.group_max;
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.ht, ht * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.rep, cap * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.cnt, cap * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.slot_of, cap * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.group_rep, G * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.counters, 4 * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.grp_idx, cap * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.ung_idx, cap * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.slots, cap * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.jbases, G * SBV_GTAB_WINDOWS * (size_t)40 * sizeof(u32)));   // 40 dwords = one Jacobian base (p256_group.h)
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.bases, G * SBV_GTAB_WINDOWS * (size_t)2 * sizeof(sbv::apt)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.jstate, G * (size_t)27 * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.gacc, 36 * c.cap * sizeof(u32)));       // limb-major with the scratch's stride; 36 words (P-256: XYZZ, 9-limb coordinates) or 32 (Ed25519) per tuple
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.ktab, G * (size_t)SBV_KEYTAB_ENTRIES * sizeof(sbv::apt)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.kvalid, G));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.tmp, G * SBV_GTAB_WINDOWS * (size_t)(SBV_GTAB_PER_WINDOW * 32) * sizeof(u32)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&b.acc, cap));
    b.ht_mask = (u32)(ht - 1);
    b.max_groups = (u32)G;
    b.min_count = c.group_min_count;
    b.cap = cap;
    b.gacc_cap = c.cap;
    return SBV_OK;
}

int ensure_ed_group_buffers(Context& c, size_t n) {
    int rc = ensure_group_buffers(c, n);
    if (rc != SBV_OK) return rc;
    sbv::EdGroupBuffers& e = c.edgrp;
    if (e.cap >= c.grp.cap && e.max_groups == c.grp.max_groups) return SBV_OK;
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    if (e.ktab) (void)hipFree(e.ktab);
    if (e.okb) (void)hipFree(e.okb);
    e = sbv::EdGroupBuffers();
    HIP_TRY(SBV_ENOMEM, hipMalloc(&e.ktab, (size_t)c.grp.max_groups * SBV_ED_KEYTAB_ENTRIES_PER_KEY * sizeof(sbv::aniels)));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&e.okb, c.grp.cap));
    e.cap = c.grp.cap;
    e.max_groups = c.grp.max_groups;
    return SBV_OK;
}

// one chunk (n <= cap) of Ed25519 tuples on `stream`: grouped step or the one-lane kernel
int enqueue_ed25519(Context& c, const uint8_t* d_tuples, size_t n, uint8_t* d_bitmap, hipStream_t stream) {
    if (c.group_enabled && n >= c.group_min_batch) {
        const int rc = ensure_ed_group_buffers(c, n);
        if (rc != SBV_OK) return rc;
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.gsync.ev_fork, stream));
        HIP_TRY(SBV_EDEVICE, sbv::launch_ed25519_verify_grouped(d_tuples, n, c.grp, c.edgrp, c.d_qtab, c.d_btab, d_bitmap, stream, c.gsync));
        return SBV_OK;
    }
    HIP_TRY(SBV_EDEVICE, sbv::launch_ed25519_verify(d_tuples, n, c.d_qtab, c.d_btab, d_bitmap, stream));
    return SBV_OK;
}

// enqueue stage A + stage B for n <= cap tuples on `stream`
int enqueue(Context& c, const uint8_t* d_tuples, size_t n, uint8_t* d_bitmap, hipStream_t stream,
            hipEvent_t after_prep, hipEvent_t* dom = nullptr, int* dom_pairs = nullptr, bool* was_grouped = nullptr) {
    const sbv::Scratch s = scratch_view(c);
    const bool grouped = c.group_enabled && n >= c.group_min_batch;
    if (was_grouped) *was_grouped = grouped;
    if (grouped) {
        const int rc = ensure_group_buffers(c, n);
        if (rc != SBV_OK) return rc;
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.gsync.ev_fork, stream));
    }
    if (grouped) {           // stage A is enqueued by the grouped launcher, in slices pipelined with the G phase
        HIP_TRY(SBV_EDEVICE, sbv::launch_p256_verify_grouped(d_tuples, s, n, c.grp, c.d_qtab, c.d_gtab, c.d_g16r, d_bitmap, stream, c.gsync,
                                                             after_prep, dom, dom_pairs));
        return SBV_OK;
    }
    HIP_TRY(SBV_EDEVICE, sbv::launch_p256_prep(d_tuples, n, s, stream));
    if (after_prep) HIP_TRY(SBV_EDEVICE, hipEventRecord(after_prep, stream));
    if (dom) HIP_TRY(SBV_EDEVICE, hipEventRecord(dom[0], stream));     // ungrouped: the dominant kernel is all of stage B
    HIP_TRY(SBV_EDEVICE, sbv::launch_p256_verify(s, n, c.d_qtab, c.d_gtab, d_bitmap, c.d_rerun, stream));
    if (dom) { HIP_TRY(SBV_EDEVICE, hipEventRecord(dom[1], stream)); if (dom_pairs) *dom_pairs = 1; }
    return SBV_OK;
}

int enqueue_keyed(Context& c, const uint8_t* d_rsh, const u32* d_slots, size_t n, uint8_t* d_bitmap, hipStream_t stream,
                  hipEvent_t after_prep) {
    const sbv::Scratch s = scratch_view(c);
    HIP_TRY(SBV_EDEVICE, sbv::launch_p256_prep(d_rsh, n, s, stream, true));
    if (after_prep) HIP_TRY(SBV_EDEVICE, hipEventRecord(after_prep, stream));
    HIP_TRY(SBV_EDEVICE, sbv::launch_p256_verify_keyed(s, n, d_slots, (u32)c.nkeys, c.d_ktab, c.d_kvalid, c.d_gtab, d_bitmap, c.d_rerun, stream));
    return SBV_OK;
}

constexpr size_t kKeyTabBytes = (size_t)SBV_KEYTAB_ENTRIES * sizeof(sbv::apt);

int ensure_key_capacity(Context& c, size_t want) {
    if (want <= c.key_cap) return SBV_OK;
    size_t cap = c.key_cap ? c.key_cap : 16;
    while (cap < want) cap *= 2;
    sbv::apt* nt = nullptr;
    uint8_t* nv = nullptr;
    HIP_TRY(SBV_ENOMEM, hipMalloc(&nt, cap * kKeyTabBytes));
    HIP_TRY(SBV_ENOMEM, hipMalloc(&nv, cap));
    HIP_TRY(SBV_EDEVICE, hipMemset(nv, 0, cap));
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());          // nothing in flight may still read the old tables
    if (c.nkeys) {
        HIP_TRY(SBV_EDEVICE, hipMemcpy(nt, c.d_ktab, c.nkeys * kKeyTabBytes, hipMemcpyDeviceToDevice));
        HIP_TRY(SBV_EDEVICE, hipMemcpy(nv, c.d_kvalid, c.nkeys, hipMemcpyDeviceToDevice));
    }
    if (c.d_btab) (void)hipFree(c.d_btab);
    c.d_btab = nullptr;
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

extern "C" int sbv_init(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
    if (c.ready) return c.device == device ? SBV_OK : SBV_EINVAL;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_err = "no HIP device visible (libsbv has no CPU fallback)";
        return SBV_ENODEV;
    }
    if (device < 0 || device >= ndev) { g_err = "device index out of range"; return SBV_EINVAL; }
    HIP_TRY(SBV_ENODEV, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(SBV_ENODEV, hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        g_err = std::string("device is ") + prop.gcnArchName + ", libsbv is built for gfx950 only";
        return SBV_ENODEV;
    }
    HIP_TRY(SBV_ENODEV, hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    for (hipStream_t* st : {&c.gsync.side_a, &c.gsync.side_b})
        HIP_TRY(SBV_ENODEV, hipStreamCreateWithFlags(st, hipStreamNonBlocking));
    for (hipEvent_t* ev : group_events(c)) HIP_TRY(SBV_ENODEV, hipEventCreateWithFlags(ev, hipEventDisableTiming));
    c.gsync.chunks = 2;
    if (const char* e = getenv("SBV_GROUP_CHUNKS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= SBV_GROUP_MAX_CHUNKS) c.gsync.chunks = v;
    }
    if (const char* e = getenv("SBV_GROUP_PARTS")) c.gsync.parts = atoi(e);
    if (const char* e = getenv("SBV_GROUP_SLICES")) c.gsync.slices = atoi(e);
    if (const char* e = getenv("SBV_GENERIC_STREAM")) {
        if (e[0] == '1') HIP_TRY(SBV_ENODEV, hipStreamCreateWithFlags(&c.gsync.side_c, hipStreamNonBlocking));
    }
    for (auto& ev : c.ev) HIP_TRY(SBV_ENODEV, hipEventCreate(&ev));
    HIP_TRY(SBV_ENODEV, hipEventCreateWithFlags(&c.busy, hipEventDisableTiming));
    // fixed-base table: computed once on the host with the same field code, then resident in HBM
    // (16-bit comb, 35.7 MB: u1*G is 17 mixed additions; built by 17 host threads in ~0.1 s)
    const size_t gcount = SBV_G16_ENTRIES;
    std::vector<sbv::apt> h_gtab(gcount);
    sbv::host_build_g16(h_gtab.data());
    HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_gtab, gcount * sizeof(sbv::apt)));
    HIP_TRY(SBV_EDEVICE, hipMemcpy(c.d_gtab, h_gtab.data(), gcount * sizeof(sbv::apt), hipMemcpyHostToDevice));
    {
        std::vector<sbv::apt> h_g16r(gcount);
        sbv::host_convert_table_r261(h_gtab.data(), h_g16r.data(), gcount);
        HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_g16r, gcount * sizeof(sbv::apt)));
        HIP_TRY(SBV_EDEVICE, hipMemcpy(c.d_g16r, h_g16r.data(), gcount * sizeof(sbv::apt), hipMemcpyHostToDevice));
    }
    if (const char* e = getenv("SBV_GROUP")) c.group_enabled = e[0] != '0';
    c.device = device;
    c.ready = true;
    g_err.clear();
    return SBV_OK;
}

extern "C" int sbv_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
    if (!c.ready) return SBV_OK;
    (void)hipSetDevice(c.device);
    (void)hipDeviceSynchronize();
    free_buffers(c);
    free_group_buffers(c);
    if (c.d_gtab) (void)hipFree(c.d_gtab);
    c.d_gtab = nullptr;
    if (c.d_g16r) (void)hipFree(c.d_g16r);
    c.d_g16r = nullptr;
    if (c.d_btab) (void)hipFree(c.d_btab);
    c.d_btab = nullptr;
    if (c.d_msgs) (void)hipFree(c.d_msgs);
    if (c.d_sigs) (void)hipFree(c.d_sigs);
    if (c.d_moff) (void)hipFree(c.d_moff);
    if (c.d_soff) (void)hipFree(c.d_soff);
    c.d_msgs = c.d_sigs = nullptr; c.d_moff = c.d_soff = nullptr; c.msgs_cap = c.sigs_cap = c.moff_cap = c.soff_cap = 0;
    if (c.d_ktab) (void)hipFree(c.d_ktab);
    if (c.d_kvalid) (void)hipFree(c.d_kvalid);
    c.d_ktab = nullptr; c.d_kvalid = nullptr; c.key_cap = c.nkeys = 0;
    c.key_index.clear();
    for (auto& ev : c.ev) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    for (auto& ev : c.prof_events) (void)hipEventDestroy(ev);
    c.prof_events.clear();
    for (auto& ev : c.prof_dom) (void)hipEventDestroy(ev);
    c.prof_dom.clear();
    c.prof_dom_used = 0;
    c.prof_used = 0;
    if (c.busy) { (void)hipEventDestroy(c.busy); c.busy = nullptr; }
    if (c.stream) { (void)hipStreamDestroy(c.stream); c.stream = nullptr; }
    for (hipStream_t* st : {&c.gsync.side_a, &c.gsync.side_b, &c.gsync.side_c}) if (*st) { (void)hipStreamDestroy(*st); *st = nullptr; }
    for (hipEvent_t* ev : group_events(c)) if (*ev) { (void)hipEventDestroy(*ev); *ev = nullptr; }
    c.busy_valid = false;
    c.ready = false;
    c.device = -1;
    return SBV_OK;
}

extern "C" int sbv_p256_verify_batch_dev(const void* d_tuples, size_t n, void* d_bitmap, void* hip_stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!d_tuples || !d_bitmap || (reinterpret_cast<uintptr_t>(d_tuples) & 15)) {
        g_err = "null or misaligned device pointer";
        return SBV_EINVAL;
    }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.device));
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
        if (c.profiling) {
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

extern "C" int sbv_p256_verify_batch(const uint8_t* tuples, size_t n, uint8_t* accept_bitmap) {
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!tuples || !accept_bitmap) { g_err = "null pointer"; return SBV_EINVAL; }
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.device));
    int rc = ensure_capacity(c, n < kMaxChunk ? n : kMaxChunk);
    if (rc != SBV_OK) return rc;
    if (c.busy_valid) HIP_TRY(SBV_EDEVICE, hipStreamWaitEvent(c.stream, c.busy, 0));
    sbv_timing tm{};
    tm.n = n;
    for (size_t off = 0; off < n; off += kMaxChunk) {
        const size_t m = n - off < kMaxChunk ? n - off : kMaxChunk;
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[0], c.stream));
        HIP_TRY(SBV_EDEVICE, hipMemcpyAsync(c.d_tuples, tuples + off * SBV_TUPLE_BYTES, m * SBV_TUPLE_BYTES,
                                            hipMemcpyHostToDevice, c.stream));
        HIP_TRY(SBV_EDEVICE, hipEventRecord(c.ev[1], c.stream));
        rc = enqueue(c, c.d_tuples, m, c.d_bitmap, c.stream, c.ev[2]);
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
    c.busy_valid = false;   // stream is idle
    tm.total_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    c.timing = tm;
    return SBV_OK;
}

extern "C" int sbv_p256_register_keys(const uint8_t* keys, size_t m, uint32_t* slots_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (m == 0) return SBV_OK;
    if (!keys || !slots_out) { g_err = "null pointer"; return SBV_EINVAL; }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.device));
    // de-duplicate, assign slots.  The index is only extended AFTER both uploads succeeded: an entry left behind by a
    // failed call would hand its slot number to the next fresh key, and the first key's ID would then resolve to another
    // signer's comb.
    std::vector<size_t> fresh;                      // indices into keys[] that need a table
    std::unordered_map<std::string, u32> pending;
    for (size_t i = 0; i < m; ++i) {
        const std::string k((const char*)keys + 64 * i, 64);
        auto it = c.key_index.find(k);
        if (it != c.key_index.end()) { slots_out[i] = it->second; continue; }
        auto pt = pending.find(k);
        if (pt != pending.end()) { slots_out[i] = pt->second; continue; }
        const u32 slot = (u32)(c.nkeys + fresh.size());
        pending.emplace(k, slot);
        fresh.push_back(i);
        slots_out[i] = slot;
    }
    if (fresh.empty()) return SBV_OK;
    int rc = ensure_key_capacity(c, c.nkeys + fresh.size());
    if (rc != SBV_OK) return rc;
    // tables are built on the host (one-time setup, same field code as the kernels), in parallel
    std::vector<sbv::apt> tabs(fresh.size() * (size_t)SBV_KEYTAB_ENTRIES);
    std::vector<uint8_t> valid(fresh.size(), 0);
    {
        size_t nt = std::thread::hardware_concurrency();
        if (nt == 0) nt = 1;
        if (nt > 64) nt = 64;
        if (nt > fresh.size()) nt = fresh.size();
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                for (size_t j = t; j < fresh.size(); j += nt)
                    valid[j] = sbv::host_build_key_table(keys + 64 * fresh[j], &tabs[j * (size_t)SBV_KEYTAB_ENTRIES]) ? 1 : 0;
            });
        for (auto& x : th) x.join();
    }
    HIP_TRY(SBV_EDEVICE, hipMemcpy(c.d_ktab + c.nkeys * (size_t)SBV_KEYTAB_ENTRIES, tabs.data(), tabs.size() * sizeof(sbv::apt),
                                   hipMemcpyHostToDevice));
    HIP_TRY(SBV_EDEVICE, hipMemcpy(c.d_kvalid + c.nkeys, valid.data(), valid.size(), hipMemcpyHostToDevice));
    for (auto& kv : pending) c.key_index.emplace(kv.first, kv.second);
    c.nkeys += fresh.size();
    return SBV_OK;
}

extern "C" int sbv_p256_key_count(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_ctx.ready ? (int)g_ctx.nkeys : SBV_ENOTINIT;
}

extern "C" int sbv_p256_clear_keys(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
    if (!c.ready) return SBV_ENOTINIT;
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.device));
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    c.key_index.clear();
    c.nkeys = 0;
    return SBV_OK;
}

extern "C" int sbv_p256_verify_batch_keyed_dev(const void* d_rsh, const void* d_slots, size_t n, void* d_bitmap, void* hip_stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!d_rsh || !d_slots || !d_bitmap || (reinterpret_cast<uintptr_t>(d_rsh) & 15)) { g_err = "null or misaligned device pointer"; return SBV_EINVAL; }
    if (c.nkeys == 0) { g_err = "no keys registered"; return SBV_EINVAL; }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.device));
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
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!rsh || !slots || !accept_bitmap) { g_err = "null pointer"; return SBV_EINVAL; }
    if (c.nkeys == 0) { g_err = "no keys registered"; return SBV_EINVAL; }
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.device));
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
int ensure_ed_table(Context& c) {
    if (c.d_btab) return SBV_OK;
    std::vector<sbv::aniels> h(SBV_ED_B16_ENTRIES);      // 16-bit comb of B: 50 MB, built by 16 host threads in ~0.2 s
    sbv::host_build_ed_b16(h.data());
    HIP_TRY(SBV_ENOMEM, hipMalloc(&c.d_btab, h.size() * sizeof(sbv::aniels)));
    HIP_TRY(SBV_EDEVICE, hipMemcpy(c.d_btab, h.data(), h.size() * sizeof(sbv::aniels), hipMemcpyHostToDevice));
    return SBV_OK;
}
}  // namespace

extern "C" int sbv_ed25519_verify_batch_dev(const void* d_tuples, size_t n, void* d_bitmap, void* hip_stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!d_tuples || !d_bitmap || (reinterpret_cast<uintptr_t>(d_tuples) & 15)) { g_err = "null or misaligned device pointer"; return SBV_EINVAL; }
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.device));
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
        if ((rc = enqueue_ed25519(c, src + off * 128, m, dst + off / 8, stream)) != SBV_OK) {
            if (hipEventRecord(c.busy, stream) == hipSuccess) c.busy_valid = true;
            return rc;
        }
        if (end) HIP_TRY(SBV_EDEVICE, hipEventRecord(end, stream));
    }
    HIP_TRY(SBV_EDEVICE, hipEventRecord(c.busy, stream));
    c.busy_valid = true;
    return SBV_OK;
}

extern "C" int sbv_ed25519_verify_batch(const uint8_t* tuples, size_t n, uint8_t* accept_bitmap) {
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
    if (!c.ready) { g_err = "sbv_init has not succeeded"; return SBV_ENOTINIT; }
    if (n == 0) return SBV_OK;
    if (!tuples || !accept_bitmap) { g_err = "null pointer"; return SBV_EINVAL; }
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.device));
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
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
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
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.device));
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
    HIP_TRY(SBV_EDEVICE, sbv::launch_msg_frontend(c.d_msgs, c.d_moff, c.d_sigs, c.d_soff, n, reinterpret_cast<u32*>(c.d_tuples), c.stream));
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
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
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
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.device));
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

extern "C" void* sbv_host_alloc(size_t bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
    if (!c.ready || bytes == 0) return nullptr;
    if (hipSetDevice(c.device) != hipSuccess) return nullptr;
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
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
    c.group_enabled = enabled != 0;
    if (min_batch) c.group_min_batch = min_batch;
    if (min_count) c.group_min_count = min_count;
    if (max_groups) c.group_max = max_groups;
    return SBV_OK;
}

extern "C" int sbv_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_ctx.profiling = on != 0;
    return SBV_OK;
}

extern "C" int sbv_profile_read_dominant(double* dominant_us, uint64_t* dominant_launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
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
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
    if (!c.ready) return SBV_ENOTINIT;
    if (!out) return SBV_EINVAL;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!c.grp.counters) return SBV_OK;
    HIP_TRY(SBV_EDEVICE, hipSetDevice(c.device));
    HIP_TRY(SBV_EDEVICE, hipDeviceSynchronize());
    uint32_t h[4];
    HIP_TRY(SBV_EDEVICE, hipMemcpy(h, c.grp.counters, sizeof h, hipMemcpyDeviceToHost));
    out[0] = h[0] < c.grp.max_groups ? h[0] : c.grp.max_groups;
    out[1] = h[1];
    out[2] = h[2];
    out[3] = h[3];
    return SBV_OK;
}

extern "C" int sbv_profile_read(double* prep_us, double* verify_us, uint64_t* launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    Context& c = g_ctx;
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
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out) return SBV_EINVAL;
    *out = g_ctx.timing;
    return SBV_OK;
}

extern "C" const char* sbv_last_error(void) {
    // per-thread: valid until this thread's next failing libsbv call
    return g_err.c_str();
}

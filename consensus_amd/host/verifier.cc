#include "verifier.h"

#include <functional>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/sbv.h"
#include "ed25519_host.h"
#include "k256_host.h"
#include "p256_host.h"

namespace sbvhost {

namespace {
// Host-side tuple preparation (SHA-256 of Msg, DER parse, digest binding) is ~1 us per signature
// per core; for 10^4..10^6-signature batches it must not run on one thread or it — not the GPU —
// bounds the batch.  Fork-join over a persistent pool: creating threads per call costs
// more than preparing a 10 000-request proposal does.
class WorkerPool {
 public:
    static WorkerPool& get() { static WorkerPool p; return p; }
    size_t size() const { return workers_.size(); }
    // runs fn(k) for k = 0..jobs-1 on the pool and on the calling thread; returns when all are done
    void run(size_t jobs, const std::function<void(size_t)>& fn) {
        if (jobs == 0) return;
        std::unique_lock<std::mutex> call(call_mu_);          // one fork-join at a time
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn; jobs_ = jobs; next_ = 0; pending_ = jobs; ++epoch_;
        }
        cv_work_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }

 private:
    WorkerPool() {
        size_t n = std::thread::hardware_concurrency();
        if (n == 0) n = 1;
        if (n > 32) n = 32;
        for (size_t i = 0; i + 1 < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void drain() {
        for (;;) {
            size_t k;
            const std::function<void(size_t)>* fn;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (!fn_ || next_ >= jobs_) return;
                k = next_++;
                fn = fn_;
            }
            (*fn)(k);
            bool last;
            {
                std::lock_guard<std::mutex> lk(mu_);
                last = --pending_ == 0;
            }
            if (last) cv_done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return stop_ || epoch_ != seen; });
                if (stop_) return;
                seen = epoch_;
            }
            drain();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_, call_mu_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t jobs_ = 0, next_ = 0, pending_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

template <typename F>
void parallel_chunks(size_t n, F fn) {
    const size_t min_per_job = 256;
    WorkerPool& pool = WorkerPool::get();
    size_t jobs = n / min_per_job;
    const size_t max_jobs = 4 * (pool.size() + 1);
    if (jobs > max_jobs) jobs = max_jobs;
    if (jobs <= 1) { fn(0, n); return; }
    const size_t per = (n + jobs - 1) / jobs;
    const std::function<void(size_t)> job = [&](size_t k) {
        const size_t lo = k * per, hi = lo + per < n ? lo + per : n;
        if (lo < hi) fn(lo, hi);
    };
    pool.run(jobs, job);
}
}  // namespace

// ---- backends ------------------------------------------------------------------------------------
namespace {
class SbvBackend : public Backend {
 public:
    // device >= 0: that GPU.  device < 0: every GPU of the node (sbv_init_all): generic batches go through the sharded
    // entry (split over the devices above 2 x 2^18 tuples, otherwise one device round-robin); registered keys and the
    // message front end stay on the first device.
    explicit SbvBackend(int device) : all_(device < 0) {
        if (all_) { const int n = sbv_init_all(); rc_ = n > 0 ? SBV_OK : (n < 0 ? n : SBV_ENODEV); }
        else rc_ = sbv_init(device);
    }
    int verify(const uint8_t* tuples, size_t n, uint8_t* bitmap) override {
        if (rc_ != SBV_OK) return rc_;                 // no device: every batch is UNAVAILABLE, never a CPU guess
        if (all_) return sbv_p256_verify_batch_sharded(tuples, n, 0, 0, bitmap, nullptr, nullptr);
        return sbv_p256_verify_batch(tuples, n, bitmap);
    }
    long register_key(const uint8_t q[64]) override {
        if (rc_ != SBV_OK) return -1;
        uint32_t slot = 0;
        return sbv_p256_register_keys(q, 1, &slot) == SBV_OK ? (long)slot : -1;
    }
    void widen_key(long slot) override {
        if (rc_ != SBV_OK || slot < 0) return;
        const uint32_t s = (uint32_t)slot;
        (void)sbv_p256_widen_keys(&s, 1);              // best effort: without its wide comb the key keeps the 8-bit one
    }
    int verify_keyed(const uint8_t* rsh, const uint32_t* slots, size_t n, uint8_t* bitmap) override {
        if (rc_ != SBV_OK) return rc_;
        // A decision-replay batch (controller.go:587-633: 50 000 decisions x 11 signatures) takes the sharded registered-key entry:
        // every GPU of the node holds the consenters' combs, uploads run in pieces beside the kernels (also on one device); a vote
        // burst or a proposal's worth of signatures stays on the latency forms of the first device.
        if (n > 32768) return sbv_p256_verify_batch_keyed_sharded(rsh, slots, n, 0, 0, bitmap, nullptr, nullptr);
        return sbv_p256_verify_batch_keyed(rsh, slots, n, bitmap);
    }
    int verify_ed25519(const uint8_t* tuples128, size_t n, uint8_t* bitmap) override {
        if (rc_ != SBV_OK) return rc_;
        return sbv_ed25519_verify_batch(tuples128, n, bitmap);
    }
    int verify_k256(const uint8_t* tuples, size_t n, uint8_t* bitmap) override {
        if (rc_ != SBV_OK) return rc_;
        return sbv_secp256k1_verify_batch(tuples, n, bitmap);
    }
    void* host_alloc(size_t bytes) override { return rc_ == SBV_OK ? sbv_host_alloc(bytes) : nullptr; }
    void host_free(void* p) override { sbv_host_free(p); }
    int verify_msgs_keyed(const uint8_t* msgs, const uint64_t* moff, const uint8_t* sigs, const uint64_t* soff,
                          const uint32_t* slots, size_t n, uint8_t* bitmap) override {
        if (rc_ != SBV_OK) return rc_;
        // a decision-replay batch: every GPU of the node, uploads in pieces beside SHA-256 + DER + verification of the piece before
        if (n > 32768) return sbv_p256_verify_msgs_keyed_sharded(msgs, moff, sigs, soff, slots, n, 0, 0, bitmap, nullptr, nullptr);
        return sbv_p256_verify_msgs_keyed(msgs, moff, sigs, soff, slots, n, bitmap);
    }
 private:
    bool all_;
    int rc_;
};
class CallbackBackend : public Backend {
 public:
    CallbackBackend(backend_fn fn, void* user, bool registry) : fn_(fn), user_(user), registry_(registry) {}
    int verify(const uint8_t* tuples, size_t n, uint8_t* bitmap) override { return fn_(tuples, n, bitmap, user_); }
    // stand-in for the registered-key form: slots are indices into a host-side key list, verify_keyed
    // re-attaches the keys and goes through the same callback (unknown slot -> all-zero key -> reject)
    long register_key(const uint8_t q[64]) override {
        if (!registry_) return -1;
        std::lock_guard<std::mutex> lk(mu_);
        const std::string k((const char*)q, 64);
        for (size_t i = 0; i < keys_.size(); ++i) if (keys_[i] == k) return (long)i;
        keys_.push_back(k);
        return (long)keys_.size() - 1;
    }
    int verify_keyed(const uint8_t* rsh, const uint32_t* slots, size_t n, uint8_t* bitmap) override {
        if (!registry_) return -2;
        std::vector<uint8_t> tuples(n * 160, 0);
        {
            std::lock_guard<std::mutex> lk(mu_);
            ++keyed_batches_;
            for (size_t i = 0; i < n; ++i) {
                memcpy(&tuples[i * 160], rsh + i * 96, 96);
                if (slots[i] < keys_.size()) memcpy(&tuples[i * 160 + 96], keys_[slots[i]].data(), 64);
            }
        }
        return fn_(tuples.data(), n, bitmap, user_);
    }
    // stand-in for the device front end (sbv_p256_verify_msgs_keyed): SHA-256 + strict DER on the host, then the callback
    int verify_msgs_keyed(const uint8_t* msgs, const uint64_t* moff, const uint8_t* sigs, const uint64_t* soff,
                          const uint32_t* slots, size_t n, uint8_t* bitmap) override {
        if (!registry_) return -2;
        std::vector<uint8_t> tuples(n * 160, 0);
        {
            std::lock_guard<std::mutex> lk(mu_);
            ++keyed_batches_;
            for (size_t i = 0; i < n; ++i)
                if (slots[i] < keys_.size()) memcpy(&tuples[i * 160 + 96], keys_[slots[i]].data(), 64);
        }
        parallel_chunks(n, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                sbv_p256_parse_der(sigs + soff[i], (size_t)(soff[i + 1] - soff[i]), &tuples[i * 160]);   // failure leaves r = s = 0
                sha256(msgs + moff[i], (size_t)(moff[i + 1] - moff[i]), &tuples[i * 160 + 64]);
            }
        });
        return fn_(tuples.data(), n, bitmap, user_);
    }
    uint64_t keyed_batches() override { std::lock_guard<std::mutex> lk(mu_); return keyed_batches_; }
    // stand-in for sbv_p256_widen_keys: remembers which slots were named consenters' (idempotent, like the library)
    void widen_key(long slot) override {
        if (!registry_ || slot < 0) return;
        std::lock_guard<std::mutex> lk(mu_);
        if ((size_t)slot < keys_.size() && std::find(widened_.begin(), widened_.end(), slot) == widened_.end()) widened_.push_back(slot);
    }
    uint64_t widened_keys() override { std::lock_guard<std::mutex> lk(mu_); return widened_.size(); }
    // the stand-in knows which scheme its test runs: the same callback receives the 128-byte tuples
    int verify_ed25519(const uint8_t* tuples128, size_t n, uint8_t* bitmap) override { return fn_(tuples128, n, bitmap, user_); }
    int verify_k256(const uint8_t* tuples, size_t n, uint8_t* bitmap) override { return fn_(tuples, n, bitmap, user_); }
 private:
    backend_fn fn_;
    void* user_;
    bool registry_;
    std::vector<long> widened_;
    std::mutex mu_;
    std::vector<std::string> keys_;
    uint64_t keyed_batches_ = 0;
};
}  // namespace
std::shared_ptr<Backend> make_sbv_backend(int device) { return std::make_shared<SbvBackend>(device); }
std::shared_ptr<Backend> make_callback_backend(backend_fn fn, void* user, bool with_key_registry) {
    return std::make_shared<CallbackBackend>(fn, user, with_key_registry);
}

// ---- coalescer -----------------------------------------------------------------------------------
// One step of a short busy wait.  The pause alone is not enough: a submitter usually wakes its caller's next thread right
// before it starts to spin, the scheduler places the woken thread on the waker's CPU, and a spinning task is not preempted
// for a fresh wake-up — the next vote of the burst then arrives only when the spinner blocks (measured: 0.4 - 2.4 ms
// between the votes of a 4-thread burst).  Yielding lets a co-located runnable thread in; with nobody waiting it is a
// ~0.3 us system call.
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
    std::this_thread::yield();
}

Coalescer::Coalescer(std::shared_ptr<Backend> be, size_t max_batch, std::chrono::microseconds max_wait)
    : be_(be), max_batch_(max_batch ? max_batch : 1), max_wait_(max_wait) {}

Coalescer::~Coalescer() {}      // no thread of its own: a submitter never returns before its job is done

// Concurrent single-signature calls are merged by the callers themselves: the first submitter that finds no leader BECOMES
// the leader (it is awake and on a core already — a dispatcher thread would have to be woken first: 30-60 us of futex latency at
// the head of a ~100 us round trip, which is what round 3's M2 paid), polls the queue for the rest of the burst, ships the
// batch and hands the verdicts out; everybody else spins on its own job's flag.  When the leader's own job is done it steps down
// at once; if jobs that arrived during its backend call are still queued, leadership passes to one of their (spinning) owners.
int Coalescer::submit(const uint8_t tuple[160], long slot, bool ed25519, bool k256, std::string* err) {
    Job j;
    memcpy(j.tuple, tuple, ed25519 ? 128 : 160);
    j.slot = slot;
    j.ed25519 = ed25519;
    j.k256 = k256;
    bool lead = false;
    static const bool trace = [] { const char* e = getenv("SBVH_TRACE"); return e && e[0] == '1'; }();
    if (trace) j.t_push = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    {
        std::lock_guard<SpinLock> lk(mu_);
        q_.push_back(&j);
        qn_.store(q_.size(), std::memory_order_release);
        ++st_.calls;
        if (!leader_) { leader_ = true; leader_flag_.store(true, std::memory_order_release); lead = true; }
    }
    // A quorum-sized batch is back in ~100 us; a futex sleep + wake-up costs 30-60 us on each side of it.  Spin for the
    // expected round trip (the caller would otherwise have burnt ~100 us of CPU verifying on its own), then sleep.
    const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(400);
    for (;;) {
        if (lead) {
            serve_as_leader(&j.done);
            lead = false;
            if (j.done.load(std::memory_order_acquire)) break;      // always: the leader's own job is in its first batch
        }
        if (j.done.load(std::memory_order_acquire)) break;
        if (!leader_flag_.load(std::memory_order_acquire)) {        // the leader left while this job was still queued: take over
            std::lock_guard<SpinLock> lk(mu_);
            if (!leader_ && !j.done.load(std::memory_order_acquire)) { leader_ = true; leader_flag_.store(true, std::memory_order_release); lead = true; }
            continue;
        }
        if (std::chrono::steady_clock::now() > spin_until) {
            std::unique_lock<std::mutex> lk(sleep_mu_);
            sleepers_.fetch_add(1, std::memory_order_seq_cst);
            cv_done_.wait(lk, [&] { return j.done.load(std::memory_order_acquire) || !leader_flag_.load(std::memory_order_acquire); });
            sleepers_.fetch_sub(1, std::memory_order_acq_rel);
            continue;
        }
        cpu_relax();
    }
    if (err && j.result < 0) *err = j.err;
    return j.result;
}

int Coalescer::submit_many(const uint8_t* tuples, size_t n, uint8_t* bitmap) {
    {
        std::lock_guard<SpinLock> lk(mu_);
        ++st_.batches;
        if (n > st_.max_batch) st_.max_batch = n;
    }
    return be_->verify(tuples, n, bitmap);      // the backend serialises device work itself
}

int Coalescer::submit_many_ed25519(const uint8_t* tuples128, size_t n, uint8_t* bitmap) {
    {
        std::lock_guard<SpinLock> lk(mu_);
        ++st_.batches;
        if (n > st_.max_batch) st_.max_batch = n;
    }
    return be_->verify_ed25519(tuples128, n, bitmap);
}

int Coalescer::submit_many_k256(const uint8_t* tuples, size_t n, uint8_t* bitmap) {
    {
        std::lock_guard<SpinLock> lk(mu_);
        ++st_.batches;
        if (n > st_.max_batch) st_.max_batch = n;
    }
    return be_->verify_k256(tuples, n, bitmap);
}

int Coalescer::submit_many_msgs_keyed(const uint8_t* msgs, const uint64_t* moff, const uint8_t* sigs, const uint64_t* soff, const uint32_t* slots,
                                      size_t n, uint8_t* bitmap) {
    const int rc = be_->verify_msgs_keyed(msgs, moff, sigs, soff, slots, n, bitmap);
    if (rc != -2) {
        std::lock_guard<SpinLock> lk(mu_);
        ++st_.batches;
        if (n > st_.max_batch) st_.max_batch = n;
    }
    return rc;
}

int Coalescer::submit_many_keyed(const uint8_t* rsh, const uint32_t* slots, size_t n, uint8_t* bitmap) {
    {
        std::lock_guard<SpinLock> lk(mu_);
        ++st_.batches;
        if (n > st_.max_batch) st_.max_batch = n;
    }
    return be_->verify_keyed(rsh, slots, n, bitmap);
}

CoalescerStats Coalescer::stats() {
    std::lock_guard<SpinLock> lk(mu_);
    return st_;
}

// mu_ NOT held; leader_ is this thread.  Ships batches until ITS OWN job is done, then steps down — also when jobs that arrived
// during its backend call are still queued: one of their owners (each is spinning or sleeping in submit()) takes over through the
// takeover path there.  A leader that kept serving until it found the queue empty could be held inside submit() for as long as
// concurrent VerifyRequest / VerifySignature traffic kept arriving — the consensus thread verifying a commit vote would not
// return although its verdict was ready.
void Coalescer::serve_as_leader(const std::atomic<bool>* own_done) {
    // SBVH_TRACE=1: one line per backend batch on stderr (size, time spent collecting, backend call)
    static const bool trace = [] { const char* e = getenv("SBVH_TRACE"); return e && e[0] == '1'; }();
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_lead = trace ? now_us() : 0;
    std::vector<Job*> batch;
    std::vector<uint8_t> tuples, bitmap;
    bool first_batch = true;
    for (;;) {
        bool wake_sleepers = false, stepped_down = false;
        {
            // The first job (the leader's own) is here: give concurrent callers a short window to join the batch.  The window
            // is tens of microseconds — below the kernel's timer slack — so the leader polls the queue length and leaves early
            // when the expected burst is complete or nothing new has arrived for a quiet period: a quarter of the window while
            // no burst size is known, half of it when one is (the votes of a burst arrive a microsecond apart — measured,
            // profiles/r04/m2_trace_r04h.txt: a batch of 15 is complete 15-24 us after its leader started; SBVH_TRACE=1 prints every arrival — and must
            // not be cut into several serial round trips), but only a SIXTH of it (8 us) while the leader is still alone: a LONE
            // call — VerifyRequest from HandleRequest, the serial verifyPrevCommitSignatures loop of
            // internal/bft/view.go:630-644, a view-change VerifySignature — must not sit out the window (at N = 16 that was
            // 10 x 25 us per sequence).  Later batches of the same leadership (jobs that arrived during the backend call) go
            // out at once.
            if (first_batch) {
                const auto t_first = std::chrono::steady_clock::now();
                const auto deadline = t_first + max_wait_;
                const auto quiet = max_wait_ / 4;
                size_t seen = 1;
                auto last = t_first;
                for (;;) {
                    const auto now = std::chrono::steady_clock::now();
                    const size_t have = qn_.load(std::memory_order_acquire);
                    const size_t hint = burst_hint_.load(std::memory_order_relaxed);
                    if (have >= max_batch_ || (hint && have >= hint) || now >= deadline) break;
                    if (have != seen) { seen = have; last = now; }
                    else if (now - last >= (have == 1 ? max_wait_ / 6 : (hint ? 2 * quiet : quiet))) break;    // alone so far: a sixth of the window (8 us)
                    cpu_relax();
                }
            }
            first_batch = false;
            std::lock_guard<SpinLock> lk(mu_);
            batch.clear();
            while (!q_.empty() && batch.size() < max_batch_) { batch.push_back(q_.front()); q_.pop_front(); }
            qn_.store(q_.size(), std::memory_order_release);
            if (batch.empty()) {                       // nothing left: step down (a job queued from now on finds no leader)
                leader_ = false;
                leader_flag_.store(false, std::memory_order_release);
                std::atomic_thread_fence(std::memory_order_seq_cst);          // as below: flag first, then look for sleepers
                wake_sleepers = sleepers_.load(std::memory_order_seq_cst) > 0;
                stepped_down = true;
            }
            if (!stepped_down) {
                ++st_.batches;
                if (batch.size() > st_.max_batch) st_.max_batch = batch.size();
            }
        }
        if (stepped_down) {
            if (wake_sleepers) { std::lock_guard<std::mutex> lk(sleep_mu_); cv_done_.notify_all(); }
            return;
        }
        const size_t n = batch.size();
        const double t_ship = trace ? now_us() : 0;
        if (trace && n > 1) {
            std::string a;
            for (size_t i = 0; i < n; ++i) a += " " + std::to_string((int)(batch[i]->t_push - t_lead));
            fprintf(stderr, "[sbvh] coalescer arrivals (us after the leader took over):%s\n", a.c_str());
        }
        bitmap.assign((n + 7) / 8, 0);
        bool all_keyed = true;
        for (size_t i = 0; i < n; ++i) all_keyed = all_keyed && batch[i]->slot >= 0;
        int rc;
        if (batch[0]->ed25519) {         // Ed25519 Verifier: 128-byte tuples
            tuples.resize(n * 128);
            for (size_t i = 0; i < n; ++i) memcpy(&tuples[i * 128], batch[i]->tuple, 128);
            rc = be_->verify_ed25519(tuples.data(), n, bitmap.data());
        } else if (batch[0]->k256) {     // secp256k1 Verifier: the same 160-byte tuples, the other curve
            tuples.resize(n * 160);
            for (size_t i = 0; i < n; ++i) memcpy(&tuples[i * 160], batch[i]->tuple, 160);
            rc = be_->verify_k256(tuples.data(), n, bitmap.data());
        } else if (all_keyed) {          // the commit-vote burst: every signer is a registered consenter
            tuples.resize(n * 96);
            std::vector<uint32_t> slots(n);
            for (size_t i = 0; i < n; ++i) { memcpy(&tuples[i * 96], batch[i]->tuple, 96); slots[i] = (uint32_t)batch[i]->slot; }
            rc = be_->verify_keyed(tuples.data(), slots.data(), n, bitmap.data());
        } else {
            tuples.resize(n * 160);
            for (size_t i = 0; i < n; ++i) memcpy(&tuples[i * 160], batch[i]->tuple, 160);
            rc = be_->verify(tuples.data(), n, bitmap.data());
        }
        const std::string err_text = rc != 0 ? std::string(sbv_last_error()) : std::string();     // this thread made the failing call
        if (trace) fprintf(stderr, "[sbvh] coalescer batch n=%zu: %.0f us since this leader started, backend %.0f us\n", n, t_ship - t_lead, now_us() - t_ship);
        {
            std::lock_guard<SpinLock> lk(mu_);
            for (size_t i = 0; i < n; ++i) {
                Job* job = batch[i];        // not touched after done: the submitter's stack frame may be gone
                job->result = rc != 0 ? (rc < 0 ? rc : -1) : ((bitmap[i >> 3] >> (i & 7)) & 1);
                if (rc != 0) job->err = err_text;
                job->done.store(true, std::memory_order_release);
            }
        }
        // store(done) above, load(sleepers_) here, and a sleeper does the mirror image (count itself, then look at done): without
        // a full fence between the two both sides may read the old value and the sleeper would never be woken
        std::atomic_thread_fence(std::memory_order_seq_cst);
        bool wake = sleepers_.load(std::memory_order_seq_cst) > 0;
        const bool step_down = own_done && own_done->load(std::memory_order_acquire);
        if (step_down) {
            // this caller's verdict is ready: hand the queue (if anything is in it) to one of the waiting owners
            std::lock_guard<SpinLock> lk(mu_);
            leader_ = false;
            leader_flag_.store(false, std::memory_order_release);
            std::atomic_thread_fence(std::memory_order_seq_cst);          // flag first, then look for sleepers (they count themselves, then look at the flag)
            wake = wake || sleepers_.load(std::memory_order_seq_cst) > 0;
        }
        if (wake) { std::lock_guard<std::mutex> lk(sleep_mu_); cv_done_.notify_all(); }
        if (step_down) return;
    }
}

// ---- verifier ------------------------------------------------------------------------------------
Verifier::Verifier(std::shared_ptr<Backend> be, const VerifierOptions& opt)
    : opt_(opt), co_(be, opt.coalesce_max, opt.coalesce_wait) {}

Verifier::~Verifier() {
    {
        std::lock_guard<std::mutex> lk(dw_mu_);
        dw_stop_ = true;
        dw_cv_.notify_all();
    }
    if (dw_thread_.joinable()) dw_thread_.join();       // drains the queue first: a queued job's slot must become ready
    for (Staging* s : {&st_msgs_, &st_sigs_, &st_moff_, &st_soff_, &st_slots_}) staging_release(*s);
}

void Verifier::staging_release(Staging& s) {
    if (s.p) {
        if (s.pinned) co_.backend().host_free(s.p);
        else free(s.p);
    }
    s = Staging();
}

void* Verifier::staging(Staging& s, size_t bytes) {
    if (bytes == 0) bytes = 1;
    if (s.cap >= bytes) return s.p;
    staging_release(s);
    size_t cap = 4096;
    while (cap < bytes) cap *= 2;
    s.p = co_.backend().host_alloc(cap);
    s.pinned = s.p != nullptr;
    if (!s.p) s.p = malloc(cap);
    s.cap = s.p ? cap : 0;
    return s.p;
}

void Verifier::RegisterConsenter(uint64_t id, const uint8_t* q) {
    const long slot = ed() || k256() ? -1 : co_.backend().register_key(q);     // -1: no key registry (Ed25519: grouped per batch; secp256k1: no combs yet)
    if (slot >= 0) co_.backend().widen_key(slot);      // consenters sign every vote of the epoch: 16 comb additions per u2 * Q instead of 32
    bytes key((const char*)q, key_bytes());
    key.resize(64, '\0');
    std::lock_guard<SpinLock> lk(mu_);
    consenters_[id] = key;
    consenter_slot_[id] = slot;
    // A commit burst is N-1 votes (view.go:537-541).  Shipping already at Quorum-1 votes (what the View waits for, view.go:531)
    // was measured in round 4 and lost: 125-136 us to the Quorum-1-th accept against 81-88 us with one batch of N-1
    // (profiles/r04/m2_trace_r04h.txt).
    co_.set_burst_hint(consenters_.size() > 1 ? consenters_.size() - 1 : 0);
}
// Clients are a registry too (the application hands their keys to the Verifier), so their keys take the same
// registered-key slots as the consenters': VerifyRequest / VerifyProposal then run 50 table additions per
// signature instead of the 256-doubling chain of a key the device has never seen.
void Verifier::RegisterClient(const std::string& client_id, const uint8_t* q) {
    bool on_device;
    { std::lock_guard<SpinLock> lk(mu_); on_device = opt_.device_client_keys; }
    const long slot = ed() || k256() || !on_device ? -1 : co_.backend().register_key(q);     // -1: no key registry (Ed25519: grouped per batch; secp256k1: no combs yet)
    bytes key((const char*)q, key_bytes());
    key.resize(64, '\0');
    std::lock_guard<SpinLock> lk(mu_);
    clients_[client_id] = key;
    client_slot_[client_id] = slot;
}
void Verifier::SetVerificationSequence(uint64_t s) { std::lock_guard<SpinLock> lk(mu_); seq_ = s; }
uint64_t Verifier::VerificationSequence() { std::lock_guard<SpinLock> lk(mu_); return seq_; }

bool Verifier::consenter_key(uint64_t id, uint8_t q[64], long* slot) {
    std::lock_guard<SpinLock> lk(mu_);
    auto it = consenters_.find(id);
    if (it == consenters_.end()) return false;
    memcpy(q, it->second.data(), 64);
    if (slot) *slot = consenter_slot_[id];
    return true;
}
bool Verifier::client_key(const std::string& id, uint8_t q[64], long* slot) {
    std::lock_guard<SpinLock> lk(mu_);
    auto it = clients_.find(id);
    if (it == clients_.end()) return false;
    memcpy(q, it->second.data(), 64);
    if (slot) {
        auto st = client_slot_.find(id);
        *slot = st == client_slot_.end() ? -1 : st->second;
    }
    return true;
}

// r|s|hash|Qx|Qy.  A DER failure leaves r = s = 0, which the kernel's range check rejects — the
// same verdict crypto/ecdsa.VerifyASN1 gives, without a second code path.
void Verifier::make_tuple(const uint8_t q[64], const bytes& msg, const bytes& sig_der, uint8_t out[160]) {
    sbv_p256_parse_der((const uint8_t*)sig_der.data(), sig_der.size(), out);
    sha256(msg.data(), msg.size(), out + 64);
    memcpy(out + 96, q, 64);
}

// R|S|A|k.  A signature that is not exactly 64 bytes becomes S = 2^256 - 1 >= L, which the kernel rejects — the verdict
// crypto/ed25519.Verify gives for a wrong length, without a second code path.
void Verifier::make_tuple_ed25519(const uint8_t a_enc[32], const bytes& msg, const bytes& sig, uint8_t out[128]) {
    if (sig.size() != 64) {
        memset(out, 0xFF, 128);
        memcpy(out + 64, a_enc, 32);
        return;
    }
    memcpy(out, sig.data(), 64);
    memcpy(out + 64, a_enc, 32);
    ed25519_hram((const uint8_t*)sig.data(), a_enc, msg.data(), msg.size(), out + 96);
}

// Key of the verified-signature cache.  It must be INJECTIVE in (key, signature, message): with a plain
// concatenation q | sig | msg two different (sig, msg) pairs whose bytes concatenate to the same string would
// share one entry, and a verified (sig, msg) would vouch for (sig + msg[:k], msg[k:]) — a message nobody signed.
// Both variable-length fields are therefore length-prefixed (fixed 8 bytes, little-endian).
std::string Verifier::cache_key(const uint8_t q[64], const bytes& msg, const bytes& sig) const {
    bytes cat((const char*)q, 64);
    auto put_len = [&cat](size_t n) {
        char b[8];
        for (int i = 0; i < 8; ++i) b[i] = (char)((uint64_t)n >> (8 * i));
        cat.append(b, 8);
    };
    put_len(sig.size());
    cat += sig;
    put_len(msg.size());
    cat += msg;
    return sha256(cat);
}

Status Verifier::verify_one(const uint8_t q[64], const bytes& msg, const bytes& sig, long slot) {
    // format checks that need no arithmetic come BEFORE the cache: crypto/ed25519.Verify returns false for any
    // signature that is not exactly 64 bytes, whatever was verified earlier
    if (ed() && sig.size() != 64) return Status::Invalid("invalid signature (length)");
    std::string key;
    if (opt_.cache_verified) {
        key = cache_key(q, msg, sig);
        std::lock_guard<SpinLock> lk(cache_mu_);
        auto it = cache_.find(key);
        if (it != cache_.end()) return it->second ? Status::Ok() : Status::Invalid("invalid signature (cached)");
    }
    uint8_t t[160];
    if (ed()) make_tuple_ed25519(q, msg, sig, t);
    else make_tuple(q, msg, sig, t);
    std::string err;
    const int r = co_.submit(t, ed() || k256() ? -1 : slot, ed(), k256(), &err);
    if (r < 0) return Status::Unavailable("backend error: " + err);
    if (opt_.cache_verified) {
        std::lock_guard<SpinLock> lk(cache_mu_);
        if (cache_.size() > (1u << 20)) cache_.clear();
        cache_[key] = r == 1;
    }
    return r == 1 ? Status::Ok() : Status::Invalid("invalid signature");
}

Status Verifier::VerifySignature(const Signature& s) {        // viewchanger.go:598, 660, 983, 1022, 1076
    uint8_t q[64];
    long slot = -1;
    if (!consenter_key(s.id, q, &slot)) return Status::Invalid("unknown signer");
    return verify_one(q, s.msg, s.value, slot);
}

// Proposal.Digest() once per Proposal object (formats.h: ProposalDigestSlot), also under concurrency: the <= N-1 goroutines
// of View.processCommits (view.go:537-541) all ask for the digest of the same proposal at the same moment.  Whoever installs
// the slot computes (or, from VerifyProposal, hands the computation to the worker); everybody else waits for `ready`.
bytes Verifier::digest_of(const Proposal& p) {
    std::shared_ptr<ProposalDigestSlot> slot = p.digest_slot();
    if (!slot) {
        auto fresh = std::make_shared<ProposalDigestSlot>();
        bool mine = false;
        slot = p.digest_slot_install(fresh, &mine);
        if (mine) {
            bytes d = proposal_digest_raw(p);
            std::lock_guard<std::mutex> lk(fresh->mu);
            fresh->digest = std::move(d);
            fresh->ready = fresh->released = true;
            fresh->ready_flag.store(true, std::memory_order_release);
            fresh->cv.notify_all();
            return fresh->digest;
        }
    }
    if (slot->ready_flag.load(std::memory_order_acquire)) return slot->digest;      // the burst's path: no lock, the digest is immutable
    std::unique_lock<std::mutex> lk(slot->mu);
    slot->cv.wait(lk, [&] { return slot->ready; });
    return slot->digest;
}

void Verifier::digest_prefetch(const Proposal& p, std::shared_ptr<ProposalDigestSlot>* slot_out) {
    if (p.digest_slot()) return;                        // computed, or on its way
    auto fresh = std::make_shared<ProposalDigestSlot>();
    bool mine = false;
    (void)p.digest_slot_install(fresh, &mine);
    if (!mine) return;
    *slot_out = fresh;
    std::lock_guard<std::mutex> lk(dw_mu_);
    if (!dw_thread_.joinable()) dw_thread_ = std::thread([this] { digest_worker(); });
    dw_jobs_.push_back(DigestJob{&p, fresh});
    dw_cv_.notify_one();
}

void Verifier::digest_worker() {
    for (;;) {
        DigestJob job;
        {
            std::unique_lock<std::mutex> lk(dw_mu_);
            dw_cv_.wait(lk, [&] { return dw_stop_ || !dw_jobs_.empty(); });
            if (dw_jobs_.empty()) return;               // stop requested and nothing left
            job = dw_jobs_.front();
            dw_jobs_.erase(dw_jobs_.begin());
        }
        bytes m = asn1_marshal_proposal(*job.p);        // the only read of the caller's object
        {
            std::lock_guard<std::mutex> lk(job.slot->mu);
            job.slot->released = true;
            job.slot->cv.notify_all();
        }
        bytes d = sha256(m);
        std::lock_guard<std::mutex> lk(job.slot->mu);
        job.slot->digest = std::move(d);
        job.slot->ready = true;
        job.slot->ready_flag.store(true, std::memory_order_release);
        job.slot->cv.notify_all();
    }
}

Status Verifier::VerifyConsenterSig(const Signature& s, const Proposal& prop, bytes* aux) {   // view.go:631, 834
    bytes binding, a;
    if (!consenter_msg_split(s.msg, &binding, &a)) return Status::Invalid("malformed signature message");
    if (binding != digest_of(prop)) return Status::Invalid("signature message does not match proposal");
    Status st = VerifySignature(s);
    if (!st.ok()) return st;
    if (aux) *aux = a;
    return Status::Ok();
}

bytes Verifier::AuxiliaryData(const bytes& msg) {               // view.go:1029, 1071 — no verification
    bytes a;
    consenter_msg_split(msg, nullptr, &a);
    return a;
}

RequestInfo Verifier::RequestID(const bytes& raw) {
    Request r;
    RequestInfo info;
    if (request_parse(raw, &r)) { info.client_id = r.client_id; info.id = r.id; }
    return info;
}

Status Verifier::VerifyRequest(const bytes& raw, RequestInfo* info) {   // controller.go:239, :742-745
    Request r;
    if (!request_parse(raw, &r)) return Status::Invalid("malformed request");
    uint8_t q[64];
    long slot = -1;
    if (!client_key(r.client_id, q, &slot)) return Status::Invalid("unknown client");
    Status st = verify_one(q, r.signed_part, r.sig, slot);
    if (!st.ok()) return st;
    if (info) { info->client_id = r.client_id; info->id = r.id; }
    return Status::Ok();
}

std::vector<RequestInfo> Verifier::RequestsFromProposal(const Proposal& p) {   // view.go:395, 419
    std::vector<RequestInfo> out;
    std::vector<bytes> reqs;
    if (!payload_split(p.payload(), &reqs)) return out;
    for (const bytes& raw : reqs) out.push_back(RequestID(raw));
    return out;
}

// All K request signatures of the proposal in ONE backend batch (view.go:555).
Status Verifier::VerifyProposal(const Proposal& p, std::vector<RequestInfo>* requests) {
    // SBVH_TRACE=1: phase times of this call on stderr (split / host pass / backend / total)
    static const bool trace = [] { const char* e = getenv("SBVH_TRACE"); return e && e[0] == '1'; }();
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = trace ? now() : 0;
    double t_split = 0, t_pass = 0, t_backend = 0;
    // Nothing of the payload is copied: the requests are parsed where they lie (formats.h: RequestView), and what the
    // backend needs is laid out straight from there.
    std::vector<std::pair<size_t, size_t>> reqs;
    if (!payload_split_views(p.payload(), &reqs)) return Status::Invalid("malformed proposal payload");
    if (trace) t_split = now();
    if ((uint64_t)p.verification_sequence() != VerificationSequence()) return Status::Invalid("verification sequence mismatch");
    // Proposal.Digest() starts NOW on the worker thread, beside everything below (and beside the prepare round that follows):
    // the first commit vote of this proposal finds it ready instead of hashing 1.7 MB (view.go:524, 834).  This call does not
    // return before the worker has stopped reading `p` (it marshals first — 0.1-0.2 ms — and hashes its own bytes).
    std::shared_ptr<ProposalDigestSlot> prefetched;
    digest_prefetch(p, &prefetched);
    struct ReleaseWait {
        std::shared_ptr<ProposalDigestSlot>& s;
        ~ReleaseWait() {
            if (!s) return;
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv.wait(lk, [&] { return s->released; });
        }
    } release_wait{prefetched};
    const size_t n = reqs.size();
    const size_t tb = ed() ? 128 : 160;              // tuple bytes of the scheme
    std::vector<uint8_t> bitmap((n + 7) / 8, 0);
    std::vector<RequestInfo> infos(n);
    std::vector<RequestView> views(n);
    std::vector<const bytes*> keys(n, nullptr);     // every request's client key (entries of the snapshot below)
    std::atomic<int> bad(0);            // 1 = malformed request, 2 = unknown client
    std::map<std::string, bytes> clients;           // snapshot: the workers must not contend on mu_
    std::map<std::string, long> client_slots;
    {
        std::lock_guard<SpinLock> lk(mu_);
        clients = clients_;
        client_slots = client_slot_;
    }
    std::vector<uint32_t> slots(n, 0);
    std::atomic<int> unkeyed(0);        // some client has no backend key slot: the whole batch goes the generic way
    const bytes& pl = p.payload();
    parallel_chunks(n, [&](size_t lo, size_t hi) {
        bool any_unkeyed = false;       // one store per chunk: 10 000 stores to one cache line from every worker cost 0.4 ms
        for (size_t i = lo; i < hi; ++i) {
            RequestView& r = views[i];
            if (!request_parse_view(pl, reqs[i].first, reqs[i].second, &r)) { bad.store(1); return; }
            infos[i].client_id.assign(pl, r.client_off, r.client_len);
            infos[i].id.assign(pl, r.id_off, r.id_len);
            auto it = clients.find(infos[i].client_id);
            if (it == clients.end()) { bad.store(2); return; }
            keys[i] = &it->second;
            const auto st = client_slots.find(infos[i].client_id);
            if (st == client_slots.end() || st->second < 0) any_unkeyed = true;
            else slots[i] = (uint32_t)st->second;
        }
        if (any_unkeyed) unkeyed.store(1);
    });
    if (bad.load() == 1) return Status::Invalid("malformed request in proposal");
    if (bad.load() == 2) return Status::Invalid("unknown client in proposal");
    if (trace) t_pass = now();
    if (n) {
        int rc = -2;
        if (!ed() && !k256() && !unkeyed.load()) {
            // every client has a comb slot: raw signed bytes + DER signatures + slots to the device front end (SHA-256 and
            // DER parsing run on the GPU); the host only packs the bytes into page-locked staging memory
            std::lock_guard<std::mutex> staging_lock(staging_mu_);
            uint64_t* moff = (uint64_t*)staging(st_moff_, (n + 1) * sizeof(uint64_t));
            uint64_t* soff = (uint64_t*)staging(st_soff_, (n + 1) * sizeof(uint64_t));
            uint32_t* dslots = (uint32_t*)staging(st_slots_, n * sizeof(uint32_t));
            if (!moff || !soff || !dslots) return Status::Unavailable("out of host memory");
            uint64_t a = 0, b = 0;
            for (size_t i = 0; i < n; ++i) { moff[i] = a; soff[i] = b; a += views[i].signed_len; b += views[i].sig_len; }
            moff[n] = a; soff[n] = b;
            uint8_t* mbuf = (uint8_t*)staging(st_msgs_, a);
            uint8_t* sbuf = (uint8_t*)staging(st_sigs_, b);
            if (!mbuf || !sbuf) return Status::Unavailable("out of host memory");
            parallel_chunks(n, [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                    memcpy(mbuf + moff[i], pl.data() + views[i].signed_off, views[i].signed_len);
                    memcpy(sbuf + soff[i], pl.data() + views[i].sig_off, views[i].sig_len);
                    dslots[i] = slots[i];
                }
            });
            rc = co_.submit_many_msgs_keyed(mbuf, moff, sbuf, soff, dslots, n, bitmap.data());
        }
        if (rc == -2) {
            // no front end, unregistered clients or another scheme: tuples built by the host workers (SHA-256, strict DER)
            std::vector<uint8_t> tuples(n * tb);
            parallel_chunks(n, [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                    const RequestView& r = views[i];
                    const uint8_t* q = (const uint8_t*)keys[i]->data();
                    const uint8_t* msg = (const uint8_t*)pl.data() + r.signed_off;
                    const uint8_t* sig = (const uint8_t*)pl.data() + r.sig_off;
                    uint8_t* out = &tuples[i * tb];
                    if (ed()) {
                        make_tuple_ed25519(q, bytes((const char*)msg, r.signed_len), bytes((const char*)sig, r.sig_len), out);
                    } else {
                        sbv_p256_parse_der(sig, r.sig_len, out);       // a failure leaves r = s = 0: rejected by the range check
                        sha256(msg, r.signed_len, out + 64);
                        memcpy(out + 96, q, 64);
                    }
                }
            });
            if (ed()) {
                rc = co_.submit_many_ed25519(tuples.data(), n, bitmap.data());
            } else if (k256()) {
                rc = co_.submit_many_k256(tuples.data(), n, bitmap.data());
            } else if (!unkeyed.load()) {
                std::vector<uint8_t> rsh(n * 96);       // r|s|hash; the key comes from the client's slot
                for (size_t i = 0; i < n; ++i) memcpy(&rsh[i * 96], &tuples[i * 160], 96);
                rc = co_.submit_many_keyed(rsh.data(), slots.data(), n, bitmap.data());
                if (rc == -2) rc = co_.submit_many(tuples.data(), n, bitmap.data());
            } else {
                rc = co_.submit_many(tuples.data(), n, bitmap.data());
            }
        }
        if (rc != 0) return Status::Unavailable(std::string("backend error: ") + sbv_last_error());
        if (trace) t_backend = now();
        for (size_t i = 0; i < n; ++i)
            if (!((bitmap[i >> 3] >> (i & 7)) & 1)) return Status::Invalid("invalid request signature in proposal");
    }
    if (requests) requests->swap(infos);
    if (trace)
        fprintf(stderr, "[sbvh] VerifyProposal n=%zu: split %.0f us, host pass %.0f us, backend %.0f us, total %.0f us\n", n, t_split - t_start,
                t_pass - t_split, t_backend - t_pass, now() - t_start);
    return Status::Ok();
}

Status Verifier::VerifyConsenterSigBatch(const std::vector<Signature>& sigs, const std::vector<const Proposal*>& props,
                                         std::vector<uint8_t>* out) {
    const size_t n = sigs.size();
    if (props.size() != n) return Status::Invalid("size mismatch");
    // SBVH_TRACE=1: phase times of this call on stderr (host pass / byte layout / backend / total)
    static const bool trace = [] { const char* e = getenv("SBVH_TRACE"); return e && e[0] == '1'; }();
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = trace ? now() : 0;
    double t_pass1 = 0, t_layout = 0, t_backend = 0;
    std::vector<uint8_t> bitmap((n + 7) / 8, 0), pre(n, 1);
    std::map<uint64_t, bytes> keys;                 // snapshot: the workers must not contend on mu_
    std::map<uint64_t, long> key_slots;
    {
        std::lock_guard<SpinLock> lk(mu_);
        keys = consenters_;
        key_slots = consenter_slot_;
    }
    // P-256 with every consenter registered on the backend (what RegisterConsenter does over libsbv): ONE pass over the Signature
    // objects' bytes.  The two passes + serial offset loop this replaces cost twice the GPU's time (profiles/r04/
    // replay_verifier_r04q.jsonl: 8.3 ms for 550 000 signatures, 2.7 of them in the backend).
    //   raw-messages route (default): the workers check signer and binding and copy message + DER signature into page-locked
    //     staging memory; SHA-256 and the DER parse run on the devices (sbv_p256_verify_msgs_keyed_sharded: every GPU of the node,
    //     uploads in pieces beside the kernels).  Offsets come from a two-level prefix sum: per-chunk byte totals from the sizes
    //     alone (no heap block touched), a serial scan over the few dozen chunks, then each chunk numbers its own signatures.
    //   records route (SBVH_REPLAY_RECORDS=1, and for a backend without the front end): the workers also hash and parse and write
    //     96-byte records r | s | SHA-256(msg) (sbv_p256_verify_batch_keyed_sharded: 100 B per signature over PCIe instead of ~140,
    //     at ~150 ns of SHA-256 per signature on the host — the better trade only when host cores are plentiful).
    // A pre-rejected signature (unknown signer, message not bound to its proposal) travels with slot 0xFFFFFFFF: no such key, rejected.
    static const bool replay_records = [] { const char* e = getenv("SBVH_REPLAY_RECORDS"); return e && e[0] == '1'; }();
    bool all_slotted = n && !ed() && !k256() && !keys.empty();
    for (const auto& kv : keys) { const auto ks = key_slots.find(kv.first); if (ks == key_slots.end() || ks->second < 0) all_slotted = false; }
    if (all_slotted) {
        // dense id -> slot table when the ids are small (they are node numbers), else the map
        uint64_t max_id = 0;
        for (const auto& kv : key_slots) max_id = kv.first > max_id ? kv.first : max_id;
        std::vector<int64_t> slot_of_id;
        if (max_id < 4096) { slot_of_id.assign(max_id + 1, -1); for (const auto& kv : key_slots) if (keys.count(kv.first)) slot_of_id[kv.first] = kv.second; }
        auto slot_of = [&](uint64_t id) -> int64_t {
            if (!slot_of_id.empty()) return id < slot_of_id.size() ? slot_of_id[id] : -1;
            const auto ks = key_slots.find(id);
            return ks != key_slots.end() && keys.count(id) ? ks->second : -1;
        };
        // signer known and message bound to the proposal it is presented with?  (thread-local memo of the last proposal's digest)
        struct Binder {
            Verifier* v; const Proposal* last = nullptr; bytes digest;
            bool bound(const Proposal* p, const bytes& m) {
                if (!p) return false;                            // a signature presented without its proposal is bound to nothing
                if (p != last) { last = p; digest = v->digest_of(*p); }
                // digest.size() is part of the predicate: the comparison reads 32 bytes of it
                return digest.size() == 32 && m.size() >= 40 && memcmp(m.data(), "SBV1", 4) == 0 && memcmp(m.data() + 4, digest.data(), 32) == 0 && consenter_msg_split(m, nullptr, nullptr);
            }
        };
        std::lock_guard<std::mutex> staging_lock(staging_mu_);
        uint32_t* dslots = (uint32_t*)staging(st_slots_, n * sizeof(uint32_t));
        if (!dslots) return Status::Unavailable("out of host memory");
        int krc = -2;
        if (!replay_records) {
            WorkerPool& pool = WorkerPool::get();
            size_t jobs = n / 1024;
            if (jobs > 4 * (pool.size() + 1)) jobs = 4 * (pool.size() + 1);
            if (jobs == 0) jobs = 1;
            const size_t per = ((n + jobs - 1) / jobs + 7) & ~(size_t)7;       // whole bitmap bytes per chunk: the batch may be shipped in slices
            jobs = (n + per - 1) / per;
            std::vector<uint64_t> msum(jobs + 1, 0), ssum(jobs + 1, 0);
            const std::function<void(size_t)> sizes = [&](size_t k) {
                const size_t lo = k * per, hi = lo + per < n ? lo + per : n;
                uint64_t a = 0, b = 0;
                for (size_t i = lo; i < hi; ++i) { a += sigs[i].msg.size(); b += sigs[i].value.size(); }
                msum[k + 1] = a; ssum[k + 1] = b;
            };
            const double t_a = trace ? now() : 0;
            pool.run(jobs, sizes);
            const double t_b = trace ? now() : 0;
            for (size_t k = 0; k < jobs; ++k) { msum[k + 1] += msum[k]; ssum[k + 1] += ssum[k]; }
            uint64_t* moff = (uint64_t*)staging(st_moff_, (n + 1) * sizeof(uint64_t));
            uint64_t* soff = (uint64_t*)staging(st_soff_, (n + 1) * sizeof(uint64_t));
            uint8_t* mbuf = (uint8_t*)staging(st_msgs_, msum[jobs] + 1);
            uint8_t* sbuf = (uint8_t*)staging(st_sigs_, ssum[jobs] + 1);
            if (!moff || !soff || !mbuf || !sbuf) return Status::Unavailable("out of host memory");
            const std::function<void(size_t)> layout = [&](size_t k) {
                const size_t lo = k * per, hi = lo + per < n ? lo + per : n;
                uint64_t a = msum[k], b = ssum[k];
                Binder bd{this};
                for (size_t i = lo; i < hi; ++i) {
                    const Signature& sg = sigs[i];
                    if (i + 6 < hi) {       // the two heap blocks of a Signature a few iterations ahead (dependent misses otherwise)
                        __builtin_prefetch(sigs[i + 6].msg.data());
                        __builtin_prefetch(sigs[i + 6].value.data());
                        __builtin_prefetch(sigs[i + 6].value.data() + 64);
                    }
                    moff[i] = a; soff[i] = b;
                    memcpy(mbuf + a, sg.msg.data(), sg.msg.size());
                    memcpy(sbuf + b, sg.value.data(), sg.value.size());
                    a += sg.msg.size(); b += sg.value.size();
                    const int64_t slot = slot_of(sg.id);
                    const bool good = slot >= 0 && bd.bound(props[i], sg.msg);
                    if (!good) pre[i] = 0;
                    dslots[i] = good ? (uint32_t)slot : 0xFFFFFFFFu;
                }
            };
            const double t_c = trace ? now() : 0;
            moff[n] = msum[jobs]; soff[n] = ssum[jobs];
            // (Shipping the batch in 2 or 4 slices, each by a helper thread while the workers lay out the next one, was measured in round 5
            // and lost: 4.07 ms in one call against 4.19 / 4.47 — a backend call has ~0.3 ms of fixed cost, more than the overlap wins.)
            pool.run(jobs, layout);
            if (trace) fprintf(stderr, "[sbvh trace] replay layout: setup %.0f us, sizes %.0f us, staging %.0f us, copy + binding %.0f us (%zu chunks)\n", t_a - t_start, t_b - t_a, t_c - t_b, now() - t_c, jobs);
            if (trace) t_pass1 = t_layout = now();
            krc = co_.submit_many_msgs_keyed(mbuf, moff, sbuf, soff, dslots, n, bitmap.data());
            if (trace) t_backend = now();
        }
        if (krc == -2) {
            uint8_t* rsh = (uint8_t*)staging(st_msgs_, n * 96);
            if (!rsh) return Status::Unavailable("out of host memory");
            parallel_chunks(n, [&](size_t lo, size_t hi) {
                Binder bd{this};
                for (size_t i = lo; i < hi; ++i) {
                    const Signature& sg = sigs[i];
                    const int64_t slot = slot_of(sg.id);
                    uint8_t* rec = rsh + i * 96;
                    if (slot < 0 || !bd.bound(props[i], sg.msg)) { pre[i] = 0; memset(rec, 0, 96); dslots[i] = 0xFFFFFFFFu; continue; }
                    sbv_p256_parse_der((const uint8_t*)sg.value.data(), sg.value.size(), rec);              // a parse failure leaves r = s = 0
                    sha256(sg.msg.data(), sg.msg.size(), rec + 64);
                    dslots[i] = (uint32_t)slot;
                }
            });
            if (trace) t_pass1 = t_layout = now();
            krc = co_.submit_many_keyed(rsh, dslots, n, bitmap.data());
            if (trace) t_backend = now();
        }
        if (krc != -2) {
            if (krc != 0) return Status::Unavailable(std::string("backend error: ") + sbv_last_error());
            out->resize(n);
            uint8_t* o = out->data();
            parallel_chunks(n, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) o[i] = pre[i] && ((bitmap[i >> 3] >> (i & 7)) & 1); });
            if (trace)
                fprintf(stderr, "[sbvh trace] VerifyConsenterSigBatch n=%zu (registered keys, %s): host pass %.0f us, backend %.0f us, total %.0f us\n", n,
                        replay_records ? "records" : "raw messages", t_pass1 - t_start, t_backend - t_layout, now() - t_start);
            return Status::Ok();
        }
        std::fill(pre.begin(), pre.end(), 1);       // a backend with neither entry: the general route below
    }
    std::vector<uint32_t> slots(n, 0);
    std::atomic<int> unkeyed(0);
    // pass 1 (host, parallel): signer known? message bound to its proposal?  No hashing of messages here.
    parallel_chunks(n, [&](size_t lo, size_t hi) {
        const Proposal* last = nullptr;
        bytes last_digest;
        bool any_unkeyed = false;           // one store per chunk (VerifyProposal says why)
        for (size_t i = lo; i < hi; ++i) {
            if (props[i] != last) { last = props[i]; last_digest = digest_of(*last); }
            const bytes& m = sigs[i].msg;
            const auto it = keys.find(sigs[i].id);
            const bool bound = m.size() >= 40 && m.compare(0, 4, "SBV1") == 0 && m.compare(4, 32, last_digest) == 0 &&
                               consenter_msg_split(m, nullptr, nullptr);
            if (it == keys.end() || !bound) { pre[i] = 0; continue; }
            const auto ks = key_slots.find(sigs[i].id);
            const long slot = ks == key_slots.end() ? -1 : ks->second;
            if (slot < 0) any_unkeyed = true; else slots[i] = (uint32_t)slot;
        }
        if (any_unkeyed) unkeyed.store(1);
    });
    if (trace) t_pass1 = now();
    int rc = -2;
    if (n && ed()) {
        // Ed25519: k = SHA-512(R | A | msg) mod L on the host workers, one batch of 128-byte tuples
        std::vector<uint8_t> tuples(n * 128, 0xFF);       // pre-rejected entries keep S = 2^256 - 1 >= L: rejected
        parallel_chunks(n, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i)
                if (pre[i]) make_tuple_ed25519((const uint8_t*)keys.find(sigs[i].id)->second.data(), sigs[i].msg, sigs[i].value, &tuples[i * 128]);
        });
        rc = co_.submit_many_ed25519(tuples.data(), n, bitmap.data());
        if (rc == -2) return Status::Unavailable("backend has no Ed25519 entry");
    } else if (n && k256()) {
        // secp256k1: tuples built on the host workers (SHA-256 + strict DER), one batch through the curve's entry
        std::vector<uint8_t> tuples(n * 160, 0);          // pre-rejected entries stay all-zero: rejected by the range check
        parallel_chunks(n, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i)
                if (pre[i]) make_tuple((const uint8_t*)keys.find(sigs[i].id)->second.data(), sigs[i].msg, sigs[i].value, &tuples[i * 160]);
        });
        rc = co_.submit_many_k256(tuples.data(), n, bitmap.data());
        if (rc == -2) return Status::Unavailable("backend has no secp256k1 entry");
    } else if (n && !unkeyed.load()) {
        // device front end: the host only lays the bytes out; SHA-256 and DER parsing run on the GPU.
        // Pre-rejected entries get an empty signature (DER failure -> r = s = 0 -> reject).
        // All five arrays live in grow-only staging memory, page-locked when the backend has it (sbv_host_alloc):
        // nothing is zero-filled, the workers write their own slices, and the H2D copies are plain DMA.
        std::lock_guard<std::mutex> staging_lock(staging_mu_);
        uint64_t* moff = (uint64_t*)staging(st_moff_, (n + 1) * sizeof(uint64_t));
        uint64_t* soff = (uint64_t*)staging(st_soff_, (n + 1) * sizeof(uint64_t));
        uint32_t* dslots = (uint32_t*)staging(st_slots_, n * sizeof(uint32_t));
        if (!moff || !soff || !dslots) return Status::Unavailable("out of host memory");
        uint64_t a = 0, b = 0;
        for (size_t i = 0; i < n; ++i) {
            moff[i] = a; soff[i] = b;
            a += sigs[i].msg.size();
            b += pre[i] ? sigs[i].value.size() : 0;
        }
        moff[n] = a; soff[n] = b;
        uint8_t* mbuf = (uint8_t*)staging(st_msgs_, a);
        uint8_t* sbuf = (uint8_t*)staging(st_sigs_, b);
        if (!mbuf || !sbuf) return Status::Unavailable("out of host memory");
        parallel_chunks(n, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                memcpy(mbuf + moff[i], sigs[i].msg.data(), sigs[i].msg.size());
                if (pre[i]) memcpy(sbuf + soff[i], sigs[i].value.data(), sigs[i].value.size());
                dslots[i] = slots[i];
            }
        });
        if (trace) t_layout = now();
        rc = co_.backend().verify_msgs_keyed(mbuf, moff, sbuf, soff, dslots, n, bitmap.data());
        if (trace) t_backend = now();
    }
    if (n && rc == -2 && !ed() && !k256()) {
        // backend without the front end (or unregistered signers): build tuples on the host
        std::vector<uint8_t> tuples(n * 160, 0);
        parallel_chunks(n, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                if (!pre[i]) continue;             // tuple stays all-zero: rejected by the range check
                make_tuple((const uint8_t*)keys.find(sigs[i].id)->second.data(), sigs[i].msg, sigs[i].value, &tuples[i * 160]);
            }
        });
        if (!unkeyed.load()) {
            std::vector<uint8_t> rsh(n * 96);
            parallel_chunks(n, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) memcpy(&rsh[i * 96], &tuples[i * 160], 96); });
            rc = co_.submit_many_keyed(rsh.data(), slots.data(), n, bitmap.data());
            if (rc == -2) rc = co_.submit_many(tuples.data(), n, bitmap.data());
        } else {
            rc = co_.submit_many(tuples.data(), n, bitmap.data());
        }
    }
    if (n && rc != 0) return Status::Unavailable(std::string("backend error: ") + sbv_last_error());
    out->assign(n, 0);
    for (size_t i = 0; i < n; ++i) (*out)[i] = pre[i] && ((bitmap[i >> 3] >> (i & 7)) & 1);
    if (trace)
        fprintf(stderr, "[sbvh trace] VerifyConsenterSigBatch n=%zu: host pass %.0f us, layout %.0f us, backend %.0f us, total %.0f us\n", n,
                t_pass1 - t_start, t_layout ? t_layout - t_pass1 : 0.0, t_backend ? t_backend - t_layout : 0.0, now() - t_start);
    return Status::Ok();
}

// ---- signer --------------------------------------------------------------------------------------
Signer::Signer(uint64_t id, const uint8_t private_key[32], Scheme scheme) : id_(id), scheme_(scheme) {
    memcpy(d_, private_key, 32);
    memset(q_, 0, 64);
    if (scheme_ == Scheme::ED25519) ed25519_public_key(d_, q_);
    else if (scheme_ == Scheme::SECP256K1) k256_pubkey_from_private(d_, q_);
    else pubkey_from_private(d_, q_);
}
bytes Signer::Sign(const bytes& msg) {
    if (scheme_ == Scheme::ED25519) {
        uint8_t sig[64];
        ed25519_sign(d_, msg.data(), msg.size(), sig);
        return bytes((const char*)sig, 64);
    }
    uint8_t h[32], rs[64];
    sha256(msg.data(), msg.size(), h);
    if (!(scheme_ == Scheme::SECP256K1 ? k256_sign_rfc6979(d_, h, rs) : sign_rfc6979(d_, h, rs))) return bytes();
    return der_encode_sig(rs);
}
Signature Signer::SignProposal(const Proposal& proposal, const bytes& auxiliary_input) {    // view.go:481
    Signature s;
    s.id = id_;
    s.msg = consenter_msg(proposal, auxiliary_input);
    s.value = Sign(s.msg);
    return s;
}

void compute_quorum(uint64_t n, int* q, int* f) {
    const int ff = ((int)n - 1) / 3;
    if (f) *f = ff;
    if (q) *q = (int)(((int)n + ff + 1 + 1) / 2);     // ceil((n + f + 1) / 2)
}

}  // namespace sbvhost

// gpu_object_placement.cpp — host-side mirror of the reference's ObjectPlacement interface
// (include/rio_gpu_object_placement.h) on top of the dense C ABI.  Pure host C++: interning of
// (struct_name, object_id) and "ip:port" strings, the malformed-record rule of the policy, reference
// counting for Clone, and a combining front-end for the single-object calls (concurrent callers share one device
// round trip).  Every placement fact lives in HBM; this file never mirrors the assignment.
//
// Mirrors (relative to /root/reference):
//   LocalObjectPlacement              rio-rs/src/object_placement/local.rs:12-68
//   Service::get_or_create_placement  rio-rs/src/service.rs:193-254
//
// Which rows are objects is a DEVICE fact too (RIO_GP_CFG_ROW_LIFECYCLE): a key that was never inserted, was removed
// (local.rs:36-37,60-68) or was dropped by clean_server (local.rs:51-58) has no entry in the reference's map; here its
// row carries the affinity RIO_GP_AFF_INACTIVE, which keeps it out of every whole-table solve (rio_op_tick), and the
// table's row count on the device is the high-water mark of the rows handed out, not max_objects.  Keys of such rows
// are reclaimed lazily, when the table runs full (reclaim()).
//
// Locks (always taken in this order): mu — compound operations and every device call sequence; imu — the interning
// tables, a reader-writer lock: calls whose key and address are already interned (every lookup, every sticky request) take it
// shared.  Single-object calls that need the device publish themselves on a lock-free list; whoever gets mu next serves
// everything published (flat combining, run_combined).  What the device knows of the host tables (node table, row count)
// is brought up to date under mu by sync_device() BEFORE any device call that may carry a new id, by whichever thread
// issues that call: an id can therefore never reach the device ahead of the table entry it refers to, whoever interned it.
//
// Host shadow of the assignment column (Shadow below).  The reference calls lookup / get_or_create_placement once per request
// from one task per connection (server.rs:292-304), and LocalObjectPlacement answers a hit from a hash map in ~100 ns
// (local.rs:42-49); a device round trip per call is 8-11 us however well it is shared.  So every answer the device gives
// for a row is remembered on the host — node + a stamp — and a later lookup / sticky request of that row is answered from
// there, under the shared table lock, without the device.  The shadow is a cache of DEVICE decisions, never a decision of
// its own: first touches, evictions, capacity and ticks all run on the GPU, and every call that changes rows it cannot
// name (clean_server, a tick, a request batch that cleaned a dead node, reclaimed keys) invalidates by stamp.
// rio_op_cfg.flags & RIO_OP_CFG_NO_HOST_SHADOW switches it off (A/B runs: examples/c_host_threads.c measures both).
#include <linux/futex.h>
#include <sched.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rio_gpu_object_placement.h"

namespace {

// one spin-wait step that tells the core (and its sibling thread) that this is one
#if defined(__x86_64__) || defined(__i386__)
static inline void cpu_relax() { __builtin_ia32_pause(); }
#elif defined(__aarch64__)
static inline void cpu_relax() { __asm__ __volatile__("yield" ::: "memory"); }
#else
static inline void cpu_relax() { __asm__ __volatile__("" ::: "memory"); }
#endif
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ != __ORDER_LITTLE_ENDIAN__
#error "Req::futex_word takes the upper half of a 64-bit word as the word at offset 4: little-endian hosts only"
#endif

constexpr int kFull = -1000;  // internal: the object table has no free row (reclaim, then retry)
constexpr int kNoop = -1001;  // internal: nothing to do on the device (an unknown key looked up / removed / deleted)
constexpr int kUpgrade = -1002;  // internal: the call has to change the interning tables — again, under the exclusive lock
constexpr int kHit = -1003;      // internal: answered from the host shadow, no device round trip

// Text of the calling thread's last failed call (rio_op_last_error): a failure is reported to the thread that made
// the call, so its text is that thread's too — no lock, no race with other callers' failures.
thread_local std::string t_err;
// length of the address the calling thread's last rio_op_lookup / rio_op_get_or_create_placement produced
// (rio_op_last_address_len): what a caller whose buffer was too small (RIO_GP_ERANGE) allocates before it asks again
thread_local size_t t_addr_len = 0;
// rio_op_snapshot's arrays: copies, owned by the calling thread until its next snapshot
thread_local std::vector<std::string> t_snap_store;
thread_local std::vector<const char*> t_snap_ty, t_snap_id, t_snap_addr;
thread_local std::vector<size_t> t_snap_tylen, t_snap_idlen;

// One single-object call waiting for its device round trip (see run_combined).
// One single-object call waiting for its device round trip (see run_combined).  The struct is a cache line of its own: its
// caller spins on `result` while every other waiter spins on theirs.  The serving thread reads the inputs once and answers
// with ONE store — node, flag, return code and the "done" bit packed into a word: with the answer spread over four fields the
// waiter's polling pulled the line back between the server's stores, four or five transfers per request, 40 us of a 64-request
// batch (measured).
struct alignas(64) Req {
    int kind;                 // 0 lookup | 1 get_or_create_placement | 2 update | 3 remove
    uint32_t row, req;        // dense ids (req: requester node for kind 1, new node or NONE for kind 2)
    std::atomic<uint64_t> result{0};  // kDone | rc << 40 | flag << 32 | node: stored LAST, and once, by the serving thread (the
                                      // request lives on its caller's stack)
    std::string err;          // text of ITS failure (a request that fails does not fail its batch-mates): written before `result`
    // the caller's view of the answer, unpacked by the caller itself
    uint32_t node = RIO_GP_NONE, flag = 0;
    int rc = RIO_GP_OK;
    static constexpr uint64_t kDone = 1ull << 63;
    static constexpr uint64_t kSleep = 1ull << 62;  // set by the caller before it sleeps on the word's upper half (a futex)
    uint32_t* futex_word() { return reinterpret_cast<uint32_t*>(&result) + 1; }  // little-endian: bits 32..63
    static uint64_t pack(int rc, uint32_t node, uint32_t flag) {
        return kDone | ((uint64_t)(uint8_t)rc << 40) | ((uint64_t)(flag & 0xFFu) << 32) | node;
    }
    void unpack(uint64_t w) { node = (uint32_t)w; flag = (uint32_t)(w >> 32) & 0xFFu; rc = (int)((w >> 40) & 0xFFu); }
};

// Where a caller leaves its request for the serving thread: 32 bytes, filled by the caller (in parallel with every other caller),
// read by the server as an array — a linked list through the callers' own stack frames cost the server a dependent cache miss
// per request (25 us for 28 requests, measured at 64 callers over two sockets).
constexpr uint32_t kSlots = 512;
struct alignas(32) Slot {
    std::atomic<uint64_t> tag{0};  // (generation + 1) << 8 | kind: stored LAST by the caller; the server waits for its generation's tag
    uint32_t row = 0, req = 0;
    Req* r = nullptr;
};

// Host shadow of the assignment column: entry = stamp << 16 | node (0xFFFF = not placed), one atomic u64 per row, in chunks of
// 65 536 rows allocated as rows are handed out.  Writers hold State::mu (the order of the device calls IS the order of the
// writes); readers hold nothing but the shared table lock that keeps their row id theirs.
//   valid(entry) = stamp >= base  &&  (node == none || stamp >= clean_stamp[node])
//   clean_server(j): clean_stamp[j] = ++clock  (the rows that sat on j are gone; every other row's entry stays good)
//   tick / a request batch that ran into a dead node / anything that moves rows it cannot name: base = ++clock
struct Shadow {
    static constexpr uint32_t kBits = 16, kNoneNode = 0xFFFFu;
    bool enabled = true;
    size_t nchunks = 0;
    std::unique_ptr<std::atomic<std::atomic<uint64_t>*>[]> chunks;
    std::unique_ptr<std::atomic<uint64_t>[]> clean_stamp;
    std::atomic<uint64_t> clock{1}, base{1};
    void init(uint64_t max_objects, uint32_t max_nodes, bool on) {
        enabled = on;
        nchunks = (size_t)((max_objects + (1u << kBits) - 1) >> kBits);
        chunks.reset(new std::atomic<std::atomic<uint64_t>*>[nchunks ? nchunks : 1]);
        for (size_t c = 0; c < (nchunks ? nchunks : 1); ++c) chunks[c].store(nullptr, std::memory_order_relaxed);
        clean_stamp.reset(new std::atomic<uint64_t>[max_nodes ? max_nodes : 1]);
        for (uint32_t j = 0; j < (max_nodes ? max_nodes : 1); ++j) clean_stamp[j].store(0, std::memory_order_relaxed);
    }
    ~Shadow() {
        for (size_t c = 0; c < nchunks; ++c) delete[] chunks[c].load(std::memory_order_relaxed);
    }
    bool get(uint32_t row, uint32_t* node) const {  // (any thread; the caller's row id is pinned by the shared table lock)
        if (!enabled) return false;
        const std::atomic<uint64_t>* ch = chunks[row >> kBits].load(std::memory_order_acquire);
        if (!ch) return false;
        const uint64_t e = ch[row & ((1u << kBits) - 1)].load(std::memory_order_acquire);
        const uint64_t stamp = e >> 16;
        const uint32_t nd = (uint32_t)(e & 0xFFFFu);
        if (stamp < base.load(std::memory_order_acquire)) return false;
        if (nd != kNoneNode && stamp < clean_stamp[nd].load(std::memory_order_acquire)) return false;
        *node = nd == kNoneNode ? RIO_GP_NONE : nd;
        return true;
    }
    void put(uint32_t row, uint32_t node) {  // mu held: what the device just said (or was just told) about the row
        if (!enabled) return;
        std::atomic<uint64_t>* ch = chunks[row >> kBits].load(std::memory_order_relaxed);
        if (!ch) {
            ch = new std::atomic<uint64_t>[1u << kBits];
            for (uint32_t k = 0; k < (1u << kBits); ++k) ch[k].store(0, std::memory_order_relaxed);
            chunks[row >> kBits].store(ch, std::memory_order_release);
        }
        const uint64_t nd = node == RIO_GP_NONE ? kNoneNode : (node & 0xFFFFu);
        ch[row & ((1u << kBits) - 1)].store((clock.load(std::memory_order_relaxed) << 16) | nd, std::memory_order_release);
    }
    void erase(uint32_t row) {  // mu + exclusive table lock held: the row changes hands (reclaim)
        std::atomic<uint64_t>* ch = chunks[row >> kBits].load(std::memory_order_relaxed);
        if (ch) ch[row & ((1u << kBits) - 1)].store(0, std::memory_order_release);
    }
    void clean(uint32_t node) { clean_stamp[node].store(clock.fetch_add(1, std::memory_order_acq_rel) + 1, std::memory_order_release); }
    void invalidate_all() { base.store(clock.fetch_add(1, std::memory_order_acq_rel) + 1, std::memory_order_release); }
};

// The table lock: readers are every single-object call (a shadow hit holds it for ~0.3 us and touches nothing else that is
// shared), writers are rare (a key or an address nobody has seen, a reclaim, the batched calls' interning).  std::shared_mutex
// keeps ONE reader count: two read-modify-writes of one cache line per call, 16 callers on two sockets spent most of a hit
// waiting for that line (1.6e6 hits/s from one thread, 4.4e6 from 16).  Here every thread counts itself in on a line of its
// own group (32 groups, threads dealt round robin); a writer raises its flag, then waits for every group to drain (Dekker's
// handshake: both sides sequentially consistent).  Writer-preferring: a reader that sees the flag steps back until it is gone.
class TableLock {
    static constexpr uint32_t kGroups = 32;
    struct alignas(64) Group { std::atomic<uint32_t> readers{0}; };
    Group groups_[kGroups];
    alignas(64) std::atomic<uint32_t> writer_{0};
    std::mutex wmu_;  // one writer at a time; writers queue here, asleep
    static uint32_t my_group() {
        static std::atomic<uint32_t> next{0};
        static thread_local uint32_t mine = next.fetch_add(1, std::memory_order_relaxed) % kGroups;
        return mine;
    }
    static void backoff(unsigned spin) {
        if (spin < 256) { cpu_relax(); return; }
        if (spin < 512) { sched_yield(); return; }
        const timespec ts{0, 50000};  // a reclaim or a big batch's interning holds the lock for milliseconds: do not burn a CPU on it
        nanosleep(&ts, nullptr);
    }

 public:
    void lock_shared() {
        std::atomic<uint32_t>& r = groups_[my_group()].readers;
        for (unsigned spin = 0;;) {
            r.fetch_add(1, std::memory_order_seq_cst);
            if (!writer_.load(std::memory_order_seq_cst)) return;
            r.fetch_sub(1, std::memory_order_seq_cst);
            while (writer_.load(std::memory_order_acquire)) backoff(spin++);
        }
    }
    bool try_lock_shared() {  // never waits: false while a writer holds or wants the lock
        std::atomic<uint32_t>& r = groups_[my_group()].readers;
        r.fetch_add(1, std::memory_order_seq_cst);
        if (!writer_.load(std::memory_order_seq_cst)) return true;
        r.fetch_sub(1, std::memory_order_seq_cst);
        return false;
    }
    void unlock_shared() { groups_[my_group()].readers.fetch_sub(1, std::memory_order_release); }
    void lock() {
        wmu_.lock();
        writer_.store(1, std::memory_order_seq_cst);
        for (Group& g : groups_)
            for (unsigned spin = 0; g.readers.load(std::memory_order_seq_cst) != 0;) backoff(spin++);
    }
    void unlock() {
        writer_.store(0, std::memory_order_release);
        wmu_.unlock();
    }
};

struct State {
    // (the words many threads hammer at once sit on cache lines of their own: 64 waiters trying the device lock on the line the
    //  publishers' list head and the table lock's reader count live on cost 50 us per combined batch, measured)
    alignas(64) std::mutex mu;    // compound operations and their device call sequences (taken first)
    alignas(64) std::atomic<int> busy{0};  // somebody holds mu: waiters look at this (a shared read) before they try the lock
    alignas(64) TableLock imu;          // the interning tables below (taken second, or alone by the single-object calls, which must
                            // be able to intern and publish while the serving thread waits for the device); shared: read-only use
    // Published single-object calls: tickets of the current generation (generation << 32 | requests so far) and two slot arrays
    // (generation & 1).  A caller takes a ticket, fills its slot and tags it; ticket 0 of a generation serves that generation.
    alignas(64) std::atomic<uint64_t> ticket{0};
    alignas(64) Slot slots[2][kSlots];
    alignas(64) std::vector<Req*> batch;  // (under mu) the requests the serving thread took off the list, oldest first
    std::vector<uint64_t> results;        // (under mu) their packed answers
    std::vector<int> sv_kind;             // (under mu) serve()'s scratch: no allocation per batch
    std::vector<uint32_t> sv_row, sv_req;
    struct KindBuf { std::vector<uint32_t> rows, reqs, res, fl, who; } sv_kb[4];  // the batch split by kind
    Shadow shadow;
    uint64_t dev_batches = 0, dev_requests = 0;  // (under mu) device round trips of combined batches / requests they carried
    size_t last_batch = 0, prev_batch = 0;       // (under mu) requests of the last two combined batches
    uint32_t collect_ns = 0;                     // rio_op_cfg.collect_ns
    uint32_t cpus = 1;                           // CPUs this process may use at once (affinity mask, cgroup quota)
    std::atomic<uint32_t> spin_ns{30000};        // how long a waiting caller spins before it sleeps (set by the servers)
    bool self_assign = true;                     // requests first-touch their requester whatever membership says (the default)
    rio_gp_t* gp = nullptr;
    uint64_t max_objects = 0;
    uint32_t max_nodes = 0;
    // --- under imu ---
    std::unordered_map<std::string, uint32_t> rows;   // "{type}.{id}" -> dense row (local.rs:26-29)
    std::deque<std::pair<std::string, std::string>> row_key;  // row -> the (struct_name, object_id) it is interned as
    std::vector<uint8_t> row_live;                    // row currently has a key
    std::vector<uint8_t> row_keep;                    // key created by rio_op_set_object_load and not used since: its load is
                                                      // what the caller set, so reclaim() must not recycle (and reset) the row
    std::vector<uint32_t> free_rows;                  // reclaimed rows, ready for new keys
    std::atomic<uint64_t> hi_rows{0};                 // rows ever handed out: every row id is < hi_rows (atomic: sync_device looks
                                                      // at it and at the two versions below without the table lock)
    std::unordered_map<std::string, uint32_t> nodes;  // address -> node id
    std::deque<std::string> node_addr;                // deque: element addresses are stable (rio_op_node_address)
    std::vector<uint8_t> node_alive, node_malformed;
    std::vector<uint64_t> node_cap;
    uint32_t n_malformed = 0;
    std::atomic<uint64_t> node_version{1};            // bumped on every change of the node table
    std::atomic<uint64_t> shape_version{1};           // bumped when a node is added or a capacity changes (not on liveness flips)
    bool reclaiming = false;                          // single-object calls wait (rcv) while keys are being reclaimed
    std::condition_variable_any rcv;
    // --- under mu ---
    uint64_t pushed_version = 0;                      // node_version the device holds
    uint64_t pushed_shape = 0;
    uint32_t pushed_nodes = 0;
    uint64_t pushed_rows = 0;                         // row count the device holds (rio_gp_set_num_objects)
    // ---
    std::atomic<int> inflight{0};                     // single-object calls between their intern and their return
    std::atomic<int> refs{1};
};

// A key part as the caller handed it over: NUL-terminated (len < 0) or with its length (the _n entry points: a Rust
// String may hold a NUL byte, service_object.rs:19-26).
struct Part {
    const char* p;
    ptrdiff_t len;
    Part(const char* c) : p(c ? c : ""), len(-1) {}
    Part(const char* c, size_t n) : p(c ? c : ""), len((ptrdiff_t)(c ? n : 0)) {}
    Part(const std::string& str) : p(str.data()), len((ptrdiff_t)str.size()) {}
    std::string str() const { return len < 0 ? std::string(p) : std::string(p, (size_t)len); }
};
// the keys of a batched call: NUL-terminated arrays, or arrays with lengths (the _batch_n entry points)
struct Keys {
    const char* const* tys; const size_t* tyl;
    const char* const* ids; const size_t* idl;
    Part ty(uint64_t k) const { return tyl ? Part(tys[k], tyl[k]) : Part(tys[k]); }
    Part id(uint64_t k) const { return idl ? Part(ids[k], idl[k]) : Part(ids[k]); }
};
std::string key_of(const Part& ty, const Part& id) {  // local.rs:26-29,43,61
    std::string k = ty.str();
    k += '.';
    k += id.str();
    return k;
}

// service.rs:204-213: splitn(2, ":"), a record is bad when ip or port is empty
bool malformed(const std::string& a) {
    const size_t c = a.find(':');
    return c == std::string::npos || c == 0 || c + 1 >= a.size();
}

int fail(int rc, const std::string& m) {
    t_err = m;
    return rc;
}
// mu for a compound call: taken blocking, and marked busy so that single-object callers wait on the flag, not on the lock word
struct DevLock {
    State* s;
    explicit DevLock(State* st) : s(st) { s->mu.lock(); s->busy.store(1, std::memory_order_relaxed); }
    ~DevLock() { s->busy.store(0, std::memory_order_release); s->mu.unlock(); }
    DevLock(const DevLock&) = delete;
    DevLock& operator=(const DevLock&) = delete;
};
int gp_fail(State* s, int rc) {  // only directly after the failing rio_gp_* call, under mu (nobody else calls the handle)
    const char* e = rio_gp_last_error(s->gp);
    t_err = e ? e : "";
    return rc;
}

// Bring the device's copy of the host tables up to date: the node table (any new address, liveness or capacity change)
// and the row count.  Requires mu; imu is taken here unless the caller already holds it.
int sync_device(State* s, bool imu_held) {
    // (the common case — nothing new since the last device call — takes no lock: with 64 callers interning, the serving thread
    //  queued 20 us per batch for a reader slot of the table lock here)
    if (s->node_version.load(std::memory_order_acquire) == s->pushed_version &&
        s->hi_rows.load(std::memory_order_acquire) == s->pushed_rows) return RIO_GP_OK;
    std::vector<uint64_t> cap;
    std::vector<uint8_t> alive;
    uint64_t version, shape, nrows;
    uint32_t m;
    {
        std::shared_lock<TableLock> li(s->imu, std::defer_lock);
        if (!imu_held) li.lock();
        version = s->node_version;
        shape = s->shape_version;
        m = (uint32_t)s->node_addr.size();
        nrows = s->hi_rows;
        if (version != s->pushed_version) {
            cap = s->node_cap;
            alive = s->node_alive;
        }
    }
    if (version != s->pushed_version) {
        // a liveness flip alone is one small asynchronous kernel (MembershipStorage::set_is_active pushed, not polled);
        // a new address or a capacity change replaces the node table
        const int rc = (shape == s->pushed_shape && m == s->pushed_nodes && m > 0)
                           ? rio_gp_set_alive_all(s->gp, m, alive.data())
                           : rio_gp_set_nodes(s->gp, m, cap.data(), alive.data());
        if (rc) return gp_fail(s, rc);
        s->pushed_version = version;
        s->pushed_shape = shape;
        s->pushed_nodes = m;
    }
    if (nrows != s->pushed_rows) {
        const int rc = rio_gp_set_num_objects(s->gp, nrows);
        if (rc) return gp_fail(s, rc);
        s->pushed_rows = nrows;
    }
    return RIO_GP_OK;
}

// find or create the node id of an address (imu held; excl: exclusively — a shared holder that would have to create gets kUpgrade)
int intern_node(State* s, const std::string& addr, bool create, uint32_t* out, bool mark_up = false, bool excl = true) {
    auto it = s->nodes.find(addr);
    if (it != s->nodes.end()) {
        *out = it->second;
        return RIO_GP_OK;
    }
    if (!create) {
        *out = RIO_GP_NONE;
        return RIO_GP_OK;
    }
    if (!excl) return kUpgrade;
    if (s->node_addr.size() >= s->max_nodes) return fail(RIO_GP_EINVAL, "node table full (max_nodes)");
    const uint32_t id = (uint32_t)s->node_addr.size();
    s->nodes.emplace(addr, id);
    s->node_addr.push_back(addr);
    // not a member until rio_op_set_member says so (is_active == false) — except a server that answers requests itself
    s->node_alive.push_back(mark_up ? 1 : 0);
    s->node_cap.push_back(RIO_GP_CAP_INF);
    const bool bad = malformed(addr);
    s->node_malformed.push_back(bad ? 1 : 0);
    s->n_malformed += bad;
    ++s->node_version;
    ++s->shape_version;
    *out = id;
    return RIO_GP_OK;
}

// find or create the row of a key (imu held).  kFull: no free row — the caller releases its locks, runs reclaim() and retries.
// use: the call makes (or unmakes) an object of the key — update, get_or_create_placement, remove; a key whose row is
// held for a load set ahead of its first use (row_keep) is an ordinary key from then on
int intern_row(State* s, const Part& ty, const Part& id, bool create, uint32_t* out, bool use = false, bool excl = true) {
    const std::string key = key_of(ty, id);
    auto it = s->rows.find(key);
    if (it != s->rows.end()) {
        *out = it->second;
        if (use && s->row_keep[it->second]) {
            if (!excl) return kUpgrade;
            s->row_keep[it->second] = 0;
        }
        return RIO_GP_OK;
    }
    if (!create) {
        *out = RIO_GP_NONE;
        return RIO_GP_OK;
    }
    if (!excl) return kUpgrade;
    uint32_t row;
    if (!s->free_rows.empty()) {
        row = s->free_rows.back();
        s->free_rows.pop_back();
    } else if (s->hi_rows < s->max_objects) {
        row = (uint32_t)s->hi_rows.fetch_add(1, std::memory_order_acq_rel);
        s->row_key.emplace_back();
        s->row_live.push_back(0);
        s->row_keep.push_back(0);
    } else {
        return kFull;
    }
    s->rows.emplace(key, row);
    s->row_key[row] = std::make_pair(ty.str(), id.str());
    s->row_live[row] = 1;
    s->row_keep[row] = use ? 0 : 1;  // created without a use: rio_op_set_object_load ahead of the first update / request
    *out = row;
    return RIO_GP_OK;
}

// The table is full: give the rows of keys that are no longer objects (unplaced AND affinity RIO_GP_AFF_INACTIVE on the
// device: removed, deleted, dropped by clean_server and not touched since) back to the free list and forget their keys —
// what HashMap::remove / retain do at once in the reference (local.rs:36-37,51-68).  Row ids live in single-object calls
// between their intern and their return, so those are drained first and held off meanwhile; compound calls are excluded
// by mu.  EINVAL when every row belongs to a live object.
int reclaim(State* s) {
    {
        std::unique_lock<TableLock> li(s->imu);
        while (s->reclaiming) s->rcv.wait(li);  // someone else is at it: wait, then let the caller retry
        if (!s->free_rows.empty() || s->hi_rows < s->max_objects) return RIO_GP_OK;
        s->reclaiming = true;
    }
    while (s->inflight.load(std::memory_order_acquire) != 0) sched_yield();
    int rc = RIO_GP_OK;
    size_t got = 0;
    {
        DevLock g(s);
        std::lock_guard<TableLock> gi(s->imu);
        const uint64_t n = s->hi_rows;
        std::vector<uint32_t> assign(n ? n : 1), aff(n ? n : 1), gone, ones;
        if ((rc = sync_device(s, true)) == RIO_GP_OK) {
            if ((rc = rio_gp_get_assign(s->gp, n, assign.data())) || (rc = rio_gp_get_objects(s->gp, n, nullptr, aff.data())))
                rc = gp_fail(s, rc);
        }
        if (rc == RIO_GP_OK) {
            for (uint64_t r = 0; r < n; ++r)
                if (s->row_live[r] && !s->row_keep[r] && assign[r] == RIO_GP_NONE && aff[r] == RIO_GP_AFF_INACTIVE) {
                    s->rows.erase(key_of(s->row_key[r].first, s->row_key[r].second));
                    s->row_key[r] = std::pair<std::string, std::string>();
                    s->row_live[r] = 0;
                    s->shadow.erase((uint32_t)r);  // the row changes hands: what the shadow knew of it belonged to the old key
                    s->free_rows.push_back((uint32_t)r);
                    gone.push_back((uint32_t)r);
                }
            got = gone.size();
            if (got) {  // a recycled row starts like a fresh one: load 1
                ones.assign(got, 1u);
                if ((rc = rio_gp_set_object_attrs(s->gp, got, gone.data(), ones.data(), nullptr))) rc = gp_fail(s, rc);
            }
        }
        s->reclaiming = false;
    }
    s->rcv.notify_all();
    if (rc) return rc;
    if (!got) return fail(RIO_GP_EINVAL, "object table full (max_objects live objects)");
    return RIO_GP_OK;
}

// The address into the caller's buffer, whole or not at all: the reference returns an owned String of any length
// (local.rs:42-49), so nothing is ever truncated.  RIO_GP_ERANGE: the buffer is too small — out holds "" and
// rio_op_last_address_len() says how many bytes (without the NUL) the address has.
int copy_out(const std::string& v, char* out, size_t cap) {
    t_addr_len = v.size();
    if (out && cap) out[0] = 0;
    if (!out || cap < v.size() + 1)
        return fail(RIO_GP_ERANGE, "address of " + std::to_string(v.size()) + " bytes does not fit the output buffer");
    memcpy(out, v.data(), v.size());
    out[v.size()] = 0;
    return RIO_GP_OK;
}

// the whole policy for a batch of (row, requester) pairs; rows/reqs are dense ids (mu held, device tables in sync)
int policy_batch(State* s, std::vector<uint32_t>& rows, std::vector<uint32_t>& reqs, uint32_t* out_node,
                 uint32_t* out_flag, bool tables_locked = true) {
    const uint64_t n = rows.size();
    bool any_malformed;
    {
        std::shared_lock<TableLock> li(s->imu, std::defer_lock);
        if (!tables_locked) li.lock();
        any_malformed = s->n_malformed != 0;
    }
    if (any_malformed) {
        // service.rs:213-223: a record whose address has no ip or no port is removed (only that record)
        std::vector<uint32_t> cur(n), bad;
        int rc = rio_gp_lookup_batch(s->gp, n, rows.data(), cur.data());
        if (rc) return gp_fail(s, rc);
        {
            std::shared_lock<TableLock> li(s->imu, std::defer_lock);
            if (!tables_locked) li.lock();
            for (uint64_t k = 0; k < n; ++k)
                if (cur[k] != RIO_GP_NONE && s->node_malformed[cur[k]]) bad.push_back(rows[k]);
        }
        if (!bad.empty() && (rc = rio_gp_remove_batch(s->gp, bad.size(), bad.data()))) return gp_fail(s, rc);
    }
    int rc = rio_gp_place_pending(s->gp, n, rows.data(), reqs.data(), out_node, out_flag);
    if (rc) return gp_fail(s, rc);
    // the shadow follows: a request that found its object on a dead node had that node cleaned (every object of it: rows this
    // batch does not name), so everything older than this batch is dropped first; then every requested row is where the
    // device just said it is
    bool cleaned = false;
    for (uint64_t k = 0; k < n && !cleaned && out_flag; ++k) cleaned = (out_flag[k] & RIO_GP_FLAG_REPLACED) != 0;
    if (cleaned || !out_flag) s->shadow.invalidate_all();
    // (a batch that cleaned a server: the device cleans first and places then — a first touch onto a requester that is not an
    //  active member survives there, where the reference's request order might wipe it again — so such answers are not cached:
    //  the next call about that row asks the device)
    std::shared_lock<TableLock> li(s->imu, std::defer_lock);
    if (cleaned && !tables_locked) li.lock();
    for (uint64_t k = 0; k < n; ++k)
        if (!cleaned || out_node[k] == RIO_GP_NONE || (out_node[k] < s->node_alive.size() && s->node_alive[out_node[k]]))
            s->shadow.put(rows[k], out_node[k]);
    return RIO_GP_OK;
}

// Combining front-end for the single-object calls that need the device (a lookup the shadow cannot answer, a first touch,
// update, remove): the reference is called from one tokio task per connection (server.rs:292-304), and one device round trip
// per call (8-11 us behind a mutex) would cap a provider at ~1e5 calls/s however many tasks call it.  Flat combining, with the
// roles settled at publication: a caller interns its key, takes a TICKET of the current generation, fills the slot the ticket
// names (32 bytes, in parallel with every other caller) and tags it.  Ticket 0 serves its generation: it waits for `mu`,
// closes the generation (later tickets belong to the next one, whose ticket 0 is already waiting for the lock), reads the
// slots as an array, makes ONE device round trip (rio_gp_mixed_batch: one launch runs the kinds that were asked for one after
// the other, one wait), answers every caller with one exchange on that caller's own cache line and releases the lock.
// Everybody else waits for its answer on a line nobody else touches — spinning at first, asleep on it after that.  Tenure is exactly one batch, so a server always returns to its own
// caller (round-2 advisor finding).  A lone caller pays exactly what it paid before; N concurrent callers share a round trip.
// Requests of one batch keep their arrival (ticket) order: the first request for an object decides, as in
// rio_gp_place_pending.  What this replaced, measured at 64 callers on the 2-socket host: round 4's queue + condition variable
// (1.1e5 lookups/s: every waiter fought for the queue's mutex), then a lock-free list with every waiter trying the device lock
// (the lock lay free for 33 us per batch, or was stormed), a linked list through the callers' stack frames (a dependent cache
// miss per request: 25 us per batch) and answers spread over four fields of a line the caller polls (40 us per batch).  Tried on
// top and dropped: the answers as ONE array + ONE "generation answered" word every caller polls, instead of a store into each
// caller's own line — no faster at 16 / 64 callers and 7x slower at 256 (every hardware thread polling one line).

// one batched device call for the requests `who` of one kind; t_err holds the text when it fails
int run_kind(State* s, int kind, std::vector<uint32_t>& rows, std::vector<uint32_t>& reqs, uint32_t* res, uint32_t* fl) {
    int rc;
    const size_t n = rows.size();
    if (kind == 0) {
        if ((rc = rio_gp_lookup_batch(s->gp, n, rows.data(), res))) gp_fail(s, rc);
        else for (size_t k = 0; k < n; ++k) s->shadow.put(rows[k], res[k]);
    } else if (kind == 1) {
        rc = policy_batch(s, rows, reqs, res, fl, false);
    } else if (kind == 2) {
        if ((rc = rio_gp_update_batch(s->gp, n, rows.data(), reqs.data()))) gp_fail(s, rc);
        else for (size_t k = 0; k < n; ++k) s->shadow.put(rows[k], reqs[k]);  // (in order: the last writer of a row wins here too)
    } else {
        if ((rc = rio_gp_remove_batch(s->gp, n, rows.data()))) gp_fail(s, rc);
        else for (size_t k = 0; k < n; ++k) s->shadow.put(rows[k], RIO_GP_NONE);
    }
    return rc;
}

// results[i] = the packed answer of batch[i] (Req::pack); error texts go into the requests themselves
void serve(State* s, std::vector<Req*>& batch, std::vector<uint64_t>& results) {  // mu held (NOT imu: callers keep interning and
                                                                                  // publishing during the round trip)
    results.assign(batch.size(), 0);
    // every id in this batch was interned before its request was published: whatever the device does not know yet of the
    // node table or the row count goes there now, ahead of the requests (the serving thread may not be the one that
    // interned the new address)
    const int src = sync_device(s, false);
    if (src) {
        for (size_t i = 0; i < batch.size(); ++i) { batch[i]->err = t_err; results[i] = Req::pack(src, RIO_GP_NONE, 0); }
        return;
    }
    ++s->dev_batches;
    s->dev_requests += batch.size();
    // (the requests' inputs were copied out of the slots by the caller: sv_kind / sv_row / sv_req)
    std::vector<int>& kinds = s->sv_kind;
    std::vector<uint32_t>&arow = s->sv_row, &areq = s->sv_req;
    State::KindBuf* kb = s->sv_kb;
    int present = 0;
    bool micro = true;
    for (int kind = 0; kind < 4; ++kind) { kb[kind].rows.clear(); kb[kind].reqs.clear(); kb[kind].who.clear(); }
    for (size_t i = 0; i < batch.size(); ++i) {
        State::KindBuf& b = kb[kinds[i]];
        b.rows.push_back(arow[i]); b.reqs.push_back(areq[i]); b.who.push_back((uint32_t)i);
    }
    for (int kind = 0; kind < 4; ++kind) {
        kb[kind].res.assign(kb[kind].who.size(), RIO_GP_NONE);
        kb[kind].fl.assign(kb[kind].who.size(), 0);
        present += !kb[kind].who.empty();
        micro = micro && kb[kind].who.size() <= 256;
    }
    auto answer = [&](int kind, int rc) {
        const State::KindBuf& b = kb[kind];
        for (size_t k = 0; k < b.who.size(); ++k) {
            if (rc) batch[b.who[k]]->err = t_err;
            results[b.who[k]] = Req::pack(rc, b.res[k], b.fl[k]);
        }
    };
    bool done[4] = {false, false, false, false};
    // writes first, in arrival order (sequential last-writer-wins, local.rs:22-40), then the reads and the policy calls:
    // a caller only returns after the batch, so any order inside it is a valid linearisation of concurrent calls
    static const int kOrder[4] = {2, 3, 0, 1};
    if (present >= 2 && micro) {
        // The callers of this generation asked for different things (a server's connections mix lookups, first touches and
        // removals): ONE device round trip for all of it — rio_gp_mixed_batch runs the four kinds in exactly this order in
        // one launch — instead of one per kind.
        bool any_malformed;
        {
            std::shared_lock<TableLock> li(s->imu);
            any_malformed = s->n_malformed != 0;  // (rare: policy_batch's own pre-pass handles those records)
        }
        if (!any_malformed) {
            rio_gp_mixed m;
            memset(&m, 0, sizeof m);
            m.struct_size = (uint32_t)sizeof m;
            m.n_update = (uint32_t)kb[2].who.size(); m.update_idx = kb[2].rows.data(); m.update_node = kb[2].reqs.data();
            m.n_remove = (uint32_t)kb[3].who.size(); m.remove_idx = kb[3].rows.data();
            m.n_lookup = (uint32_t)kb[0].who.size(); m.lookup_idx = kb[0].rows.data(); m.lookup_out = kb[0].res.data();
            m.n_place = (uint32_t)kb[1].who.size(); m.place_idx = kb[1].rows.data(); m.place_requester = kb[1].reqs.data();
            m.place_node = kb[1].res.data(); m.place_flag = kb[1].fl.data();
            if (rio_gp_mixed_batch(s->gp, &m) == RIO_GP_OK) {
                static const int kRcOf[4] = {2, 3, 0, 1};  // kind -> index into rio_gp_mixed.rc (update, remove, lookup, place)
                for (int kind : kOrder) {  // the shadow follows in the order the device applied them
                    const State::KindBuf& b = kb[kind];
                    if (b.who.empty() || m.rc[kRcOf[kind]] != RIO_GP_OK) continue;  // a refused kind changed nothing: one by one below
                    bool cleaned = false;
                    if (kind == 1) {
                        for (size_t k = 0; k < b.who.size() && !cleaned; ++k) cleaned = (b.fl[k] & RIO_GP_FLAG_REPLACED) != 0;
                        if (cleaned) s->shadow.invalidate_all();
                    }
                    std::shared_lock<TableLock> lk(s->imu, std::defer_lock);
                    if (cleaned) lk.lock();  // (policy_batch's rule: after a clean, answers on servers that are not active are not cached)
                    for (size_t k = 0; k < b.who.size(); ++k) {
                        const uint32_t v = kind == 2 ? b.reqs[k] : kind == 3 ? RIO_GP_NONE : b.res[k];
                        if (!cleaned || v == RIO_GP_NONE || (v < s->node_alive.size() && s->node_alive[v])) s->shadow.put(b.rows[k], v);
                    }
                    answer(kind, RIO_GP_OK);
                    done[kind] = true;
                }
            }
        }
    }
    for (int kind : kOrder) {
        State::KindBuf& b = kb[kind];
        if (b.who.empty() || done[kind]) continue;
        int rc = run_kind(s, kind, b.rows, b.reqs, b.res.data(), b.fl.data());
        if (rc == RIO_GP_OK || b.who.size() == 1) {
            answer(kind, rc);
            continue;
        }
        // The batched call was refused (the dense layer validates before it mutates): run the requests one by one, in
        // order, so that only the offender sees the error — its batch-mates are other callers' requests.
        std::vector<uint32_t> r1(1), q1(1);
        for (size_t k = 0; k < b.who.size(); ++k) {
            r1[0] = b.rows[k]; q1[0] = b.reqs[k];
            uint32_t nd = RIO_GP_NONE, f = 0;
            rc = run_kind(s, kind, r1, q1, &nd, &f);
            if (rc) batch[b.who[k]]->err = t_err;
            results[b.who[k]] = Req::pack(rc, nd, f);
        }
    }
}

// how long a waiter has been at it, in nanoseconds (steady clock: a pause is 10-40 ns depending on the core, so iteration
// counts say nothing about time)
static inline long long waited_ns(const std::chrono::steady_clock::time_point t0) {
    return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
}
// A device round trip is 8-15 us, and a caller waits for one to three of them: spin (a futex sleep + wake costs more than the
// wait); give the core away when it takes longer than `spin_ns` (more callers than cores; a compound call holds the device),
// sleep in earnest when it takes much longer (a snapshot, a reclaim, a big batched call).
static inline void wait_step(unsigned spin, const std::chrono::steady_clock::time_point t0, long long spin_ns) {
    if (spin < 64) { cpu_relax(); return; }
    for (int q = 0; q < 4; ++q) cpu_relax();
    if ((spin & 63u) != 0) return;
    const long long ns = waited_ns(t0);
    if (ns > 20 * spin_ns) { const timespec ts{0, 50000}; nanosleep(&ts, nullptr); }
    else if (ns > spin_ns) sched_yield();
}

static inline void futex_wait32(uint32_t* addr, uint32_t expect, long timeout_ns) {
    const timespec ts{0, timeout_ns};
    (void)syscall(SYS_futex, addr, FUTEX_WAIT_PRIVATE, expect, &ts, nullptr, 0);
}
static inline void futex_wake32(uint32_t* addr) { (void)syscall(SYS_futex, addr, FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0); }

// CPUs the process may keep busy: the affinity mask, cut by a cgroup CPU quota (v2 cpu.max, v1 cfs_quota_us / cfs_period_us)
static uint32_t usable_cpus() {
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = CPU_COUNT(&set);
    if (n < 1) n = 1;
    long long quota = -1, period = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else {
        FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
        FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fq && fp && (fscanf(fq, "%lld", &quota) != 1 || fscanf(fp, "%lld", &period) != 1)) quota = -1;
        if (fq) fclose(fq);
        if (fp) fclose(fp);
    }
    if (quota > 0 && period > 0) n = std::min<long>(n, std::max<long>(1, (long)((quota + period - 1) / period)));
    return (uint32_t)n;
}

int run_combined(State* s, Req* mine) {
    // publish: a ticket of the current generation, then the slot it names
    uint64_t gen;
    uint32_t idx;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        const uint64_t t = s->ticket.fetch_add(1, std::memory_order_acq_rel);
        gen = t >> 32;
        idx = (uint32_t)t;
        if (idx < kSlots) break;
        // the generation is full (more callers than slots): wait until its server closes it, then take a ticket of the next one
        while ((s->ticket.load(std::memory_order_acquire) >> 32) == gen) wait_step(spin++, t0, 300000);
    }
    Slot& mys = s->slots[gen & 1][idx];
    mys.row = mine->row;
    mys.req = mine->req;
    mys.r = mine;
    mys.tag.store(((gen + 1) << 8) | (uint64_t)mine->kind, std::memory_order_release);
    if (idx != 0) {
        // Somebody took ticket 0 of this generation: THAT caller serves it, our request included.  We only wait for our result,
        // on a cache line nobody else touches: no lock word, no shared flag (64 waiters trying the lock left it free for 33 us per
        // batch — everyone was asleep or yielding when it was released — and stormed it when it was not).
        // Spin for about as long as the answer usually takes, then sleep on the word itself (a futex; the server wakes exactly
        // the callers that said they sleep).  How long "usually" is depends on how many CPUs the callers may burn: 64 callers
        // spinning inside a 16-CPU quota are throttled as a group — the server with them (measured on the GPU box: 102 of 109
        // scheduler periods throttled, 4.7e5 calls/s) — so the servers shorten the spin when a batch carries more callers than
        // there are CPUs (spin_ns).
        uint64_t w;
        const long long budget = s->spin_ns.load(std::memory_order_relaxed);
        for (unsigned spin = 1;; ++spin) {
            if ((w = mine->result.load(std::memory_order_acquire)) & Req::kDone) break;
            cpu_relax();
            if ((spin & 15u) != 0 || waited_ns(t0) < budget) continue;
            w = mine->result.fetch_or(Req::kSleep, std::memory_order_acq_rel);
            if (w & Req::kDone) break;
            // (bounded: a wake-up that got lost would cost a millisecond, not the call)
            while (!((w = mine->result.load(std::memory_order_acquire)) & Req::kDone))
                futex_wait32(mine->futex_word(), (uint32_t)(Req::kSleep >> 32), 1000000);
            break;
        }
        mine->unpack(w);
        return mine->rc;
    }
    // Ticket 0: the server of everything published in this generation, from now until we close it.  (One spinning thread per
    // generation: it may spin long — the lock is handed over within a device round trip unless a compound call holds it.)
    for (unsigned spin = 0;; ++spin) {
        if (!s->busy.load(std::memory_order_relaxed) && s->mu.try_lock()) break;
        wait_step(spin, t0, 5000000);
    }
    s->busy.store(1, std::memory_order_relaxed);
    if (s->collect_ns > 1 && s->last_batch + s->prev_batch > 1) {
        // Other callers are active: the ones the last batch served are on their way back with their next request.  Closed-loop
        // callers otherwise alternate between two cohorts, each batch carrying half of them and every call waiting two round
        // trips; the server waits — at most collect_ns — until about as many callers have published as the last two batches
        // carried together: "half, then the other half" becomes "everyone" per round trip (16 callers: 2 023 batches for 32 000
        // requests instead of 3 900, 7.1e5 against 5.2e5 calls/s, measured).
        const uint32_t want = (uint32_t)(s->last_batch + s->prev_batch) - 1u;
        const auto c0 = std::chrono::steady_clock::now();
        // (wait_step: a window that outlasts its bound by much — the quota throttled the process — yields instead of burning on)
        for (unsigned spin = 0; (uint32_t)s->ticket.load(std::memory_order_relaxed) < want &&
                                std::chrono::steady_clock::now() - c0 < std::chrono::nanoseconds(s->collect_ns); ++spin)
            wait_step(spin, c0, (long long)s->collect_ns);
    }
    // close the generation: whoever takes a ticket from now on belongs to the next one (and its ticket 0 waits for this lock)
    const uint64_t closed = s->ticket.exchange((gen + 1) << 32, std::memory_order_acq_rel);
    const uint32_t n = (uint32_t)closed < kSlots ? (uint32_t)closed : kSlots;
    std::vector<Req*>& batch = s->batch;
    batch.resize(n);
    s->sv_kind.resize(n); s->sv_row.resize(n); s->sv_req.resize(n);
    Slot* sl = s->slots[gen & 1];
    for (uint32_t i = 0; i < n; ++i) {  // arrival (ticket) order; a caller between its ticket and its tag is waited for
        uint64_t tag;
        // (a publisher preempted between its ticket and its tag — 64-256 callers inside a 16-CPU quota — must get a CPU to finish:
        //  pause, then yield, then sleep, instead of burning the quantum it is waiting for)
        const auto w0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0; ((tag = sl[i].tag.load(std::memory_order_acquire)) >> 8) != gen + 1; ++spin) wait_step(spin, w0, 20000);
        s->sv_kind[i] = (int)(tag & 0xFFu);
        s->sv_row[i] = sl[i].row;
        s->sv_req[i] = sl[i].req;
        batch[i] = sl[i].r;
    }
    s->prev_batch = s->last_batch;
    s->last_batch = n;
    serve(s, batch, s->results);
    uint32_t* asleep[kSlots];
    uint32_t n_asleep = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (batch[i] == mine) { mine->unpack(s->results[i]); continue; }
        uint32_t* const fw = batch[i]->futex_word();
        // last touch of that request: it lives on its caller's stack
        if (batch[i]->result.exchange(s->results[i], std::memory_order_acq_rel) & Req::kSleep) asleep[n_asleep++] = fw;
    }
    // the next batch's callers spin about two of this batch's round trips when there is a CPU for each of them, less when not
    {
        const uint32_t callers = std::max<uint32_t>(1u, std::max<uint32_t>(n, (uint32_t)s->prev_batch));
        const uint64_t ns = 30000ull * s->cpus / callers;
        s->spin_ns.store((uint32_t)std::min<uint64_t>(30000, std::max<uint64_t>(2000, ns)), std::memory_order_relaxed);
    }
    s->busy.store(0, std::memory_order_release);
    s->mu.unlock();
    // wake the sleepers after the lock is gone: the next server is already at work (the word may belong to a later call of the
    // same thread by now — a spurious wake-up, which every sleeper tolerates)
    for (uint32_t i = 0; i < n_asleep; ++i) futex_wake32(asleep[i]);
    return mine->rc;
}

// A single-object call: intern under imu — shared first: a key and an address that are already known change nothing, and
// such a call may be answered from the host shadow right there (kHit); exclusively when something has to be created —
// (held off while keys are being reclaimed), count as in flight while its row id is on its way to the device (reclaim()
// waits for that count to drain before it forgets any key).
// `intern(excl)` fills the request and returns RIO_GP_OK, kFull, an error, kNoop = "nothing to do" (e.g. lookup of an unknown
// key), kHit = answered, or (excl == false only) kUpgrade.
template <typename F>
int single_call(State* s, Req* r, F intern) {
    for (int attempt = 0;; ++attempt) {
        int rc;
        {
            std::shared_lock<TableLock> li(s->imu);
            while (s->reclaiming) s->rcv.wait(li);
            rc = intern(false);
            if (rc == RIO_GP_OK) s->inflight.fetch_add(1, std::memory_order_acq_rel);
        }
        if (rc == kUpgrade) {
            std::unique_lock<TableLock> li(s->imu);
            while (s->reclaiming) s->rcv.wait(li);
            rc = intern(true);
            if (rc == RIO_GP_OK) s->inflight.fetch_add(1, std::memory_order_acq_rel);
        }
        if (rc == kFull && attempt == 0) {
            if ((rc = reclaim(s))) return rc;
            continue;
        }
        if (rc == kFull) return fail(RIO_GP_EINVAL, "object table full (max_objects live objects)");
        if (rc != RIO_GP_OK) return rc;
        rc = run_combined(s, r);
        s->inflight.fetch_sub(1, std::memory_order_acq_rel);  // what the caller still holds are node ids: never reclaimed
        if (rc) t_err = r->err;
        return rc;
    }
}

// compound (mu + imu for the whole call) operations: run `body`, reclaim and retry once when the table is full
template <typename F>
int compound_call(State* s, F body) {
    for (int attempt = 0;; ++attempt) {
        int rc;
        {
            DevLock g(s);
            std::lock_guard<TableLock> gi(s->imu);
            rc = body();
        }
        if (rc == kFull && attempt == 0) {
            if ((rc = reclaim(s))) return rc;
            continue;
        }
        if (rc == kFull) return fail(RIO_GP_EINVAL, "object table full (max_objects live objects)");
        return rc;
    }
}

}  // namespace

struct rio_op {
    State* s;
};

extern "C" {

int rio_op_create(const rio_op_cfg* cfg, rio_op_t** out) {
    if (out) *out = nullptr;
    if (!cfg || !out || cfg->struct_size != sizeof(rio_op_cfg) || cfg->max_objects == 0 || cfg->max_nodes == 0)
        return fail(RIO_GP_EINVAL, "rio_op_create: bad cfg");
    if (cfg->collect_ns > RIO_OP_MAX_COLLECT_NS)  // (the field was `reserved` in ABI version 1: a client that never set it)
        return fail(RIO_GP_EINVAL, "rio_op_create: collect_ns above RIO_OP_MAX_COLLECT_NS (1 ms)");
    rio_gp_cfg g;
    memset(&g, 0, sizeof g);
    g.struct_size = sizeof g;
    g.device = cfg->device;
    g.max_objects = cfg->max_objects;
    g.max_nodes = cfg->max_nodes;
    g.spill_rounds = cfg->spill_rounds;
    // reference-faithful unless the caller opts out: get_or_create_placement first-touches the requester whatever membership
    // says about it (service.rs:244-252)
    g.flags = RIO_GP_CFG_ROW_LIFECYCLE | ((cfg->flags & RIO_OP_CFG_LIVE_FIRST_TOUCH) ? 0u : RIO_GP_CFG_REF_SELF_ASSIGN);
    rio_gp_t* gp = nullptr;
    int rc = rio_gp_create(&g, &gp);
    if (rc) {
        const char* e = rio_gp_last_error(nullptr);
        return fail(rc, e ? e : "");
    }
    // every potential row is initialised (unplaced, load 1, not an object); none takes part until it is handed out
    if ((rc = rio_gp_set_objects(gp, cfg->max_objects, nullptr, nullptr)) || (rc = rio_gp_set_num_objects(gp, 0)) ||
        (rc = rio_gp_set_nodes(gp, 0, nullptr, nullptr))) {
        const char* e = rio_gp_last_error(gp);
        t_err = e ? e : "";
        rio_gp_destroy(gp);
        return rc;
    }
    State* s = new State();
    s->gp = gp;
    s->max_objects = cfg->max_objects;
    s->max_nodes = cfg->max_nodes;
    s->pushed_version = s->node_version.load();  // the empty node table is on the device
    s->pushed_shape = s->shape_version.load();
    s->shadow.init(cfg->max_objects, cfg->max_nodes, (cfg->flags & RIO_OP_CFG_NO_HOST_SHADOW) == 0);
    s->collect_ns = cfg->collect_ns ? cfg->collect_ns : RIO_OP_DEFAULT_COLLECT_NS;
    s->cpus = usable_cpus();
    s->self_assign = (cfg->flags & RIO_OP_CFG_LIVE_FIRST_TOUCH) == 0;
    *out = new rio_op{s};
    return RIO_GP_OK;
}

rio_op_t* rio_op_clone(rio_op_t* p) {
    if (!p) return nullptr;
    p->s->refs.fetch_add(1);
    return new rio_op{p->s};
}

void rio_op_release(rio_op_t* p) {
    if (!p) return;
    State* s = p->s;
    delete p;
    if (s->refs.fetch_sub(1) == 1) {
        rio_gp_destroy(s->gp);
        delete s;
    }
}

int rio_op_prepare(rio_op_t* p) { return p ? RIO_GP_OK : RIO_GP_EINVAL; }

const char* rio_op_last_error(rio_op_t*) { return t_err.c_str(); }

rio_gp_t* rio_op_dense(rio_op_t* p) { return p ? p->s->gp : nullptr; }

const char* rio_op_node_address(rio_op_t* p, uint32_t node_id) {
    if (!p) return nullptr;
    std::shared_lock<TableLock> gi(p->s->imu);
    // node_addr is a deque of strings that are never modified: the pointer stays valid for the life of the provider
    return node_id < p->s->node_addr.size() ? p->s->node_addr[node_id].c_str() : nullptr;
}

static int op_update_batch(rio_op_t* p, uint64_t n, const Keys& ks, const char* const* addrs) {
    if (!p || (n && (!ks.tys || !ks.ids || !addrs))) return RIO_GP_EINVAL;
    State* s = p->s;
    return compound_call(s, [&]() -> int {
        std::vector<uint32_t> rows, nodes;
        rows.reserve(n);
        nodes.reserve(n);
        for (uint64_t k = 0; k < n; ++k) {
            uint32_t row = RIO_GP_NONE, node = RIO_GP_NONE;
            int rc;
            if (addrs[k]) {  // Some(address): entry(key) = address
                if ((rc = intern_row(s, ks.ty(k), ks.id(k), true, &row, true))) return rc;
                if ((rc = intern_node(s, addrs[k], true, &node))) return rc;
            } else {         // None: remove(key) (local.rs:36-37) — an unknown key stays unknown
                if ((rc = intern_row(s, ks.ty(k), ks.id(k), false, &row, true))) return rc;
                if (row == RIO_GP_NONE) continue;
            }
            rows.push_back(row);
            nodes.push_back(node);
        }
        int rc;
        if ((rc = sync_device(s, true))) return rc;
        if (rows.empty()) return RIO_GP_OK;
        rc = rio_gp_update_batch(s->gp, rows.size(), rows.data(), nodes.data());
        if (rc) return gp_fail(s, rc);
        for (size_t k = 0; k < rows.size(); ++k) s->shadow.put(rows[k], nodes[k]);  // (in order: the last writer of a row wins)
        return RIO_GP_OK;
    });
}

int rio_op_update_batch(rio_op_t* p, uint64_t n, const char* const* tys, const char* const* ids, const char* const* addrs) {
    return op_update_batch(p, n, Keys{tys, nullptr, ids, nullptr}, addrs);
}
int rio_op_update_batch_n(rio_op_t* p, uint64_t n, const char* const* tys, const size_t* tyl, const char* const* ids, const size_t* idl,
                          const char* const* addrs) {
    if (n && (!tyl || !idl)) return RIO_GP_EINVAL;
    return op_update_batch(p, n, Keys{tys, tyl, ids, idl}, addrs);
}

static int op_update(rio_op_t* p, const Part& ty, const Part& id, const char* addr) {
    if (!p) return RIO_GP_EINVAL;
    State* s = p->s;
    Req r;
    r.kind = 2;
    r.req = RIO_GP_NONE;
    const int rc = single_call(s, &r, [&](bool excl) -> int {
        int rc;
        if (addr) {  // Some(address): entry(key) = address
            if ((rc = intern_row(s, ty, id, true, &r.row, true, excl))) return rc;
            return intern_node(s, addr, true, &r.req, false, excl);
        }
        // None: remove(key) (local.rs:36-37) — an unknown key stays unknown
        if ((rc = intern_row(s, ty, id, false, &r.row, true, excl))) return rc;
        return r.row == RIO_GP_NONE ? kNoop : RIO_GP_OK;
    });
    return rc == kNoop ? RIO_GP_OK : rc;
}

static int op_lookup_batch(rio_op_t* p, uint64_t n, const Keys& ks, uint32_t* out) {
    if (!p || (n && (!ks.tys || !ks.ids || !out))) return RIO_GP_EINVAL;
    State* s = p->s;
    DevLock g(s);
    std::shared_lock<TableLock> gi(s->imu);
    std::vector<uint32_t> rows, where;
    for (uint64_t k = 0; k < n; ++k) {
        uint32_t row;
        int rc = intern_row(s, ks.ty(k), ks.id(k), false, &row);
        if (rc) return rc;
        out[k] = RIO_GP_NONE;  // unknown key: Ok(None)
        if (row != RIO_GP_NONE) {
            rows.push_back(row);
            where.push_back((uint32_t)k);
        }
    }
    if (rows.empty()) return RIO_GP_OK;
    int rc;
    if ((rc = sync_device(s, true))) return rc;
    std::vector<uint32_t> res(rows.size());
    rc = rio_gp_lookup_batch(s->gp, rows.size(), rows.data(), res.data());
    if (rc) return gp_fail(s, rc);
    for (size_t q = 0; q < rows.size(); ++q) { out[where[q]] = res[q]; s->shadow.put(rows[q], res[q]); }
    return RIO_GP_OK;
}

int rio_op_lookup_batch(rio_op_t* p, uint64_t n, const char* const* tys, const char* const* ids, uint32_t* out) {
    return op_lookup_batch(p, n, Keys{tys, nullptr, ids, nullptr}, out);
}
int rio_op_lookup_batch_n(rio_op_t* p, uint64_t n, const char* const* tys, const size_t* tyl, const char* const* ids, const size_t* idl,
                          uint32_t* out) {
    if (n && (!tyl || !idl)) return RIO_GP_EINVAL;
    return op_lookup_batch(p, n, Keys{tys, tyl, ids, idl}, out);
}

int rio_op_update(rio_op_t* p, const char* ty, const char* id, const char* addr) { return op_update(p, Part(ty), Part(id), addr); }
int rio_op_update_n(rio_op_t* p, const char* ty, size_t ty_len, const char* id, size_t id_len, const char* addr) {
    return op_update(p, Part(ty, ty_len), Part(id, id_len), addr);
}

static int op_lookup(rio_op_t* p, const Part& ty, const Part& id, char* out, size_t cap, int* found) {
    if (!p || !found) return RIO_GP_EINVAL;
    State* s = p->s;
    Req r;
    r.kind = 0;
    r.req = RIO_GP_NONE;
    *found = 0;
    t_addr_len = 0;
    int hit_rc = RIO_GP_OK;
    const int rc = single_call(s, &r, [&](bool) -> int {
        const int rc = intern_row(s, ty, id, false, &r.row);
        if (rc) return rc;
        if (r.row == RIO_GP_NONE) return kNoop;  // unknown key: Ok(None), no device work
        // what the device last said about this row, if nothing has happened since that could have moved it: the answer
        // (local.rs:42-49 is a hash-map read; so is this).  The address is copied out here, under the table lock.
        if (s->shadow.get(r.row, &r.node)) {
            *found = r.node != RIO_GP_NONE;
            if (*found) hit_rc = copy_out(s->node_addr[r.node], out, cap);
            return kHit;
        }
        return RIO_GP_OK;
    });
    if (rc == kNoop) return RIO_GP_OK;
    if (rc == kHit) return hit_rc;
    if (rc) return rc;
    *found = r.node != RIO_GP_NONE;
    t_addr_len = 0;
    if (*found) {
        std::shared_lock<TableLock> gi(s->imu);
        return copy_out(s->node_addr[r.node], out, cap);  // RIO_GP_ERANGE: *found is set, nothing was copied
    }
    return RIO_GP_OK;
}

int rio_op_lookup(rio_op_t* p, const char* ty, const char* id, char* out, size_t cap, int* found) {
    return op_lookup(p, Part(ty), Part(id), out, cap, found);
}
int rio_op_lookup_n(rio_op_t* p, const char* ty, size_t ty_len, const char* id, size_t id_len, char* out, size_t cap, int* found) {
    return op_lookup(p, Part(ty, ty_len), Part(id, id_len), out, cap, found);
}

size_t rio_op_last_address_len(rio_op_t*) { return t_addr_len; }

// rio_op_try_*: the host shadow or RIO_GP_EAGAIN.  Nothing here takes State::mu, waits for the table lock (a batched call holds
// it exclusively across its device calls), interns a key or an address, or counts as in flight: a pure read of the interning
// tables and one shadow word, under the shared side of the table lock when it is free right now.
namespace {
struct TrySharedLock {
    TableLock& l;
    bool held;
    explicit TrySharedLock(TableLock& l_) : l(l_), held(l_.try_lock_shared()) {}
    ~TrySharedLock() { if (held) l.unlock_shared(); }
};
}  // namespace

int rio_op_try_lookup_n(rio_op_t* p, const char* ty, size_t ty_len, const char* id, size_t id_len, char* out, size_t cap, int* found) {
    if (!p || !found || (!ty && ty_len) || (!id && id_len)) return RIO_GP_EINVAL;
    State* s = p->s;
    *found = 0;
    t_addr_len = 0;
    TrySharedLock li(s->imu);
    if (!li.held || s->reclaiming) return RIO_GP_EAGAIN;
    const auto it = s->rows.find(key_of(Part(ty, ty_len), Part(id, id_len)));
    if (it == s->rows.end()) return RIO_GP_OK;  // a key nobody has interned: Ok(None), as rio_op_lookup says without the device
    uint32_t node;
    if (!s->shadow.get(it->second, &node)) return RIO_GP_EAGAIN;
    *found = node != RIO_GP_NONE;
    return *found ? copy_out(s->node_addr[node], out, cap) : RIO_GP_OK;
}

int rio_op_try_get_or_create_placement_n(rio_op_t* p, const char* ty, size_t ty_len, const char* id, size_t id_len,
                                         const char* self_address, char* out, size_t cap, uint32_t* flag) {
    if (!p || !self_address || (!ty && ty_len) || (!id && id_len)) return RIO_GP_EINVAL;
    State* s = p->s;
    t_addr_len = 0;
    TrySharedLock li(s->imu);
    if (!li.held || s->reclaiming) return RIO_GP_EAGAIN;
    const auto it = s->rows.find(key_of(Part(ty, ty_len), Part(id, id_len)));
    if (it == s->rows.end() || s->row_keep[it->second]) return RIO_GP_EAGAIN;  // a first touch: the device's
    const auto rq = s->nodes.find(self_address);
    if (rq == s->nodes.end()) return RIO_GP_EAGAIN;                             // a requester nobody has seen: interned by the call
    // the sticky path of service.rs:199-242, exactly as op_get_or_create answers it from the shadow
    uint32_t nd;
    if (!s->shadow.get(it->second, &nd) || nd == RIO_GP_NONE || nd >= s->node_alive.size() || !s->node_alive[nd] || s->node_malformed[nd])
        return RIO_GP_EAGAIN;
    if (flag) *flag = nd == rq->second ? RIO_GP_FLAG_LOCAL : RIO_GP_FLAG_REDIRECT;
    return copy_out(s->node_addr[nd], out, cap);
}

int rio_op_clean_server(rio_op_t* p, const char* address) {
    if (!p || !address) return RIO_GP_EINVAL;
    State* s = p->s;
    DevLock g(s);
    std::shared_lock<TableLock> gi(s->imu);
    uint32_t node;
    int rc = intern_node(s, address, false, &node);
    if (rc) return rc;
    if (node == RIO_GP_NONE) return RIO_GP_OK;  // nothing was ever placed there: retain() removes nothing
    if ((rc = sync_device(s, true))) return rc;
    rc = rio_gp_clean_server(s->gp, node, nullptr);
    if (rc) return gp_fail(s, rc);
    s->shadow.clean(node);  // whatever the shadow held for rows of that node is void; every other row's entry stands
    return RIO_GP_OK;
}

static int op_remove(rio_op_t* p, const Part& ty, const Part& id) {
    if (!p) return RIO_GP_EINVAL;
    State* s = p->s;
    Req r;
    r.kind = 3;
    r.req = RIO_GP_NONE;
    const int rc = single_call(s, &r, [&](bool excl) -> int {
        const int rc = intern_row(s, ty, id, false, &r.row, true, excl);
        if (rc) return rc;
        return r.row == RIO_GP_NONE ? kNoop : RIO_GP_OK;  // absent: no-op (local.rs:60-68)
    });
    return rc == kNoop ? RIO_GP_OK : rc;
}

int rio_op_remove(rio_op_t* p, const char* ty, const char* id) { return op_remove(p, Part(ty), Part(id)); }
int rio_op_remove_n(rio_op_t* p, const char* ty, size_t ty_len, const char* id, size_t id_len) {
    return op_remove(p, Part(ty, ty_len), Part(id, id_len));
}

int rio_op_len(rio_op_t* p, uint64_t* out) {
    if (!p || !out) return RIO_GP_EINVAL;
    State* s = p->s;
    DevLock g(s);
    int rc;
    if ((rc = sync_device(s, false))) return rc;
    rc = rio_gp_count_placed(s->gp, out);
    return rc ? gp_fail(s, rc) : RIO_GP_OK;
}

int rio_op_set_member(rio_op_t* p, const char* address, int active, uint64_t capacity) {
    if (!p || !address) return RIO_GP_EINVAL;
    State* s = p->s;
    DevLock g(s);
    std::lock_guard<TableLock> gi(s->imu);
    uint32_t node;
    int rc = intern_node(s, address, true, &node);
    if (rc) return rc;
    s->node_alive[node] = active ? 1 : 0;
    if (s->node_cap[node] != capacity) {
        s->node_cap[node] = capacity;
        ++s->shape_version;
    }
    ++s->node_version;
    return sync_device(s, true);
}

static int op_set_object_load(rio_op_t* p, const Part& ty, const Part& id, uint32_t load) {
    if (!p) return RIO_GP_EINVAL;
    State* s = p->s;
    return compound_call(s, [&]() -> int {
        uint32_t row;
        int rc = intern_row(s, ty, id, true, &row);
        if (rc) return rc;
        if ((rc = sync_device(s, true))) return rc;
        // the load only: the key does not become an object before its first update / request
        rc = rio_gp_set_object_attrs(s->gp, 1, &row, &load, nullptr);
        return rc ? gp_fail(s, rc) : RIO_GP_OK;
    });
}

int rio_op_set_object_load(rio_op_t* p, const char* ty, const char* id, uint32_t load) { return op_set_object_load(p, Part(ty), Part(id), load); }
int rio_op_set_object_load_n(rio_op_t* p, const char* ty, size_t ty_len, const char* id, size_t id_len, uint32_t load) {
    return op_set_object_load(p, Part(ty, ty_len), Part(id, id_len), load);
}

static int op_get_or_create_batch(rio_op_t* p, uint64_t n, const Keys& ks, const char* const* selfs, uint32_t* out_node,
                                  uint32_t* out_flag) {
    if (!p || (n && (!ks.tys || !ks.ids || !selfs || !out_node))) return RIO_GP_EINVAL;
    State* s = p->s;
    return compound_call(s, [&]() -> int {
        std::vector<uint32_t> rows(n), reqs(n);
        for (uint64_t k = 0; k < n; ++k) {
            int rc;
            if ((rc = intern_row(s, ks.ty(k), ks.id(k), true, &rows[k], true))) return rc;
            if ((rc = intern_node(s, selfs[k], true, &reqs[k], true))) return rc;  // a server answering requests is up
        }
        int rc;
        if ((rc = sync_device(s, true))) return rc;
        return policy_batch(s, rows, reqs, out_node, out_flag);
    });
}

int rio_op_get_or_create_placement_batch(rio_op_t* p, uint64_t n, const char* const* tys, const char* const* ids,
                                         const char* const* selfs, uint32_t* out_node, uint32_t* out_flag) {
    return op_get_or_create_batch(p, n, Keys{tys, nullptr, ids, nullptr}, selfs, out_node, out_flag);
}
int rio_op_get_or_create_placement_batch_n(rio_op_t* p, uint64_t n, const char* const* tys, const size_t* tyl, const char* const* ids,
                                           const size_t* idl, const char* const* selfs, uint32_t* out_node, uint32_t* out_flag) {
    if (n && (!tyl || !idl)) return RIO_GP_EINVAL;
    return op_get_or_create_batch(p, n, Keys{tys, tyl, ids, idl}, selfs, out_node, out_flag);
}

static int op_get_or_create(rio_op_t* p, const Part& ty, const Part& id, const char* self_address, char* out, size_t cap,
                            uint32_t* flag) {
    if (!p || !self_address) return RIO_GP_EINVAL;
    State* s = p->s;
    Req r;
    r.kind = 1;
    t_addr_len = 0;
    int hit_rc = RIO_GP_OK;
    const int rc = single_call(s, &r, [&](bool excl) -> int {
        int rc;
        if ((rc = intern_row(s, ty, id, true, &r.row, true, excl))) return rc;
        if ((rc = intern_node(s, self_address, true, &r.req, true, excl))) return rc;  // a server answering requests is up
        // The sticky path of service.rs:199-242 — lookup, the server it names is an active member, return it — from the host
        // shadow: the object is where the device last put it, nothing has happened since that could have moved it, and that
        // server is up and well-formed.  Anything else (pending, on a server that is not active, malformed) is the device's.
        uint32_t nd;
        if (s->shadow.get(r.row, &nd) && nd != RIO_GP_NONE && nd < s->node_alive.size() && s->node_alive[nd] && !s->node_malformed[nd]) {
            r.node = nd;
            r.flag = nd == r.req ? RIO_GP_FLAG_LOCAL : RIO_GP_FLAG_REDIRECT;
            if (flag) *flag = r.flag;
            hit_rc = copy_out(s->node_addr[nd], out, cap);
            return kHit;
        }
        return RIO_GP_OK;
    });
    if (rc == kHit) return hit_rc;
    if (rc) return rc;
    if (flag) *flag = r.flag;
    std::shared_lock<TableLock> gi(s->imu);
    // RIO_GP_ERANGE: the decision is made and *flag is set; the address is one rio_op_lookup away (a pure read)
    return copy_out(r.node == RIO_GP_NONE ? std::string() : s->node_addr[r.node], out, cap);
}

int rio_op_get_or_create_placement(rio_op_t* p, const char* ty, const char* id, const char* self_address, char* out,
                                   size_t cap, uint32_t* flag) {
    return op_get_or_create(p, Part(ty), Part(id), self_address, out, cap, flag);
}
int rio_op_get_or_create_placement_n(rio_op_t* p, const char* ty, size_t ty_len, const char* id, size_t id_len,
                                     const char* self_address, char* out, size_t cap, uint32_t* flag) {
    return op_get_or_create(p, Part(ty, ty_len), Part(id, id_len), self_address, out, cap, flag);
}

int rio_op_snapshot_key_lengths(rio_op_t* p, const size_t** struct_name_lens, const size_t** object_id_lens) {
    if (!p || !struct_name_lens || !object_id_lens) return RIO_GP_EINVAL;
    t_snap_tylen.clear(); t_snap_idlen.clear();
    for (size_t k = 0; k + 2 < t_snap_store.size(); k += 3) {
        t_snap_tylen.push_back(t_snap_store[k].size());
        t_snap_idlen.push_back(t_snap_store[k + 1].size());
    }
    *struct_name_lens = t_snap_tylen.data();
    *object_id_lens = t_snap_idlen.data();
    return RIO_GP_OK;
}

int rio_op_snapshot(rio_op_t* p, uint64_t* n_out, const char* const** struct_names, const char* const** object_ids,
                    const char* const** server_addresses) {
    if (!p || !n_out || !struct_names || !object_ids || !server_addresses) return RIO_GP_EINVAL;
    State* s = p->s;
    t_snap_store.clear();
    t_snap_ty.clear(); t_snap_id.clear(); t_snap_addr.clear();
    {
        DevLock g(s);
        std::shared_lock<TableLock> gi(s->imu);
        int rc;
        if ((rc = sync_device(s, true))) return rc;
        const uint64_t n = s->hi_rows;
        std::vector<uint32_t> assign(n ? n : 1);
        if ((rc = rio_gp_get_assign(s->gp, n, assign.data()))) return gp_fail(s, rc);
        // copies: keys can be reclaimed and the tables can grow as soon as the locks are released
        for (uint64_t row = 0; row < n; ++row) {
            const uint32_t nd = assign[row];
            if (!s->row_live[row] || nd == RIO_GP_NONE || nd >= s->node_addr.size()) continue;
            t_snap_store.push_back(s->row_key[row].first);
            t_snap_store.push_back(s->row_key[row].second);
            t_snap_store.push_back(s->node_addr[nd]);
        }
    }
    for (size_t k = 0; k + 2 < t_snap_store.size(); k += 3) {
        t_snap_ty.push_back(t_snap_store[k].c_str());
        t_snap_id.push_back(t_snap_store[k + 1].c_str());
        t_snap_addr.push_back(t_snap_store[k + 2].c_str());
    }
    *n_out = t_snap_ty.size();
    *struct_names = t_snap_ty.data();
    *object_ids = t_snap_id.data();
    *server_addresses = t_snap_addr.data();
    return RIO_GP_OK;
}

int rio_op_tick(rio_op_t* p, rio_gp_stats* stats) {
    if (!p) return RIO_GP_EINVAL;
    State* s = p->s;
    DevLock g(s);
    int rc;
    if ((rc = sync_device(s, false))) return rc;
    // A whole-table solve has no requester that vouches for itself: it places on servers that are active members only, also when
    // requests first-touch their requester whatever membership says (the default: service.rs:244-252) — an object evicted from a
    // dead server would otherwise claim that same server, its home, again.
    if (s->self_assign && (rc = rio_gp_set_flags(s->gp, RIO_GP_CFG_ROW_LIFECYCLE))) return gp_fail(s, rc);
    rc = rio_gp_tick(s->gp, stats);
    if (rc) gp_fail(s, rc);
    s->shadow.invalidate_all();  // a whole-table solve may move any pending or evicted row (also when it failed half-way)
    if (s->self_assign) {
        const int rc2 = rio_gp_set_flags(s->gp, RIO_GP_CFG_ROW_LIFECYCLE | RIO_GP_CFG_REF_SELF_ASSIGN);
        if (rc2 && !rc) return gp_fail(s, rc2);
    }
    return rc;
}

int rio_op_invalidate_cache(rio_op_t* p) {
    if (!p) return RIO_GP_EINVAL;
    DevLock g(p->s);
    p->s->shadow.invalidate_all();
    return RIO_GP_OK;
}

int rio_op_device_round_trips(rio_op_t* p, uint64_t* batches, uint64_t* requests) {
    if (!p) return RIO_GP_EINVAL;
    DevLock g(p->s);
    if (batches) *batches = p->s->dev_batches;
    if (requests) *requests = p->s->dev_requests;
    return RIO_GP_OK;
}

}  // extern "C"

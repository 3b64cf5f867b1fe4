// gpu_object_placement.cpp — host-side mirror of the reference's ObjectPlacement interface
// (include/rio_gpu_object_placement.h) on top of the dense C ABI.  Pure host C++: interning of
// (struct_name, object_id) and "ip:port" strings, the malformed-record rule of the policy, reference
// counting for Clone, and a combining front-end for the single-object calls (concurrent callers share one device
// round trip).  Every placement fact lives in HBM; this file never mirrors the assignment.
//
// Mirrors (relative to /root/reference):
//   LocalObjectPlacement              rio-rs/src/object_placement/local.rs:12-68
//   Service::get_or_create_placement  rio-rs/src/service.rs:193-254
#include <sched.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rio_gpu_object_placement.h"

namespace {

// One single-object call waiting for its device round trip (see run_combined).
struct Req {
    int kind;                 // 0 lookup | 1 get_or_create_placement | 2 update | 3 remove
    uint32_t row, req;        // dense ids (req: requester node for kind 1, new node or NONE for kind 2)
    uint32_t node = RIO_GP_NONE, flag = 0;
    int rc = RIO_GP_OK;
    std::atomic<int> done{0};  // set LAST by the serving thread: the request lives on its caller's stack
};

struct State {
    std::mutex mu;    // compound operations and their device call sequences (taken first)
    std::mutex imu;   // the interning tables below (taken second, or alone by the combined single-object calls, which
                      // must be able to intern and queue while the serving thread waits for the device)
    std::mutex qmu;                    // combiner: queue of single-object calls + who is serving it
    std::condition_variable qcv;
    std::vector<Req*> queue;
    bool serving = false;
    int sleepers = 0;                  // waiters that gave up spinning and sleep on qcv
    std::string err;
    rio_gp_t* gp = nullptr;
    uint64_t max_objects = 0;
    uint32_t max_nodes = 0;
    std::unordered_map<std::string, uint32_t> rows;   // "{type}.{id}" -> dense row (local.rs:26-29)
    std::vector<std::pair<std::string, std::string>> row_key;  // row -> the (struct_name, object_id) it was first interned as
    std::vector<const char*> snap_ty, snap_id, snap_addr;      // last rio_op_snapshot (pointers into row_key / node_addr)
    std::unordered_map<std::string, uint32_t> nodes;  // address -> node id
    std::vector<std::string> node_addr;
    std::vector<uint8_t> node_alive, node_malformed;
    std::vector<uint64_t> node_cap;
    uint32_t n_malformed = 0;
    std::atomic<int> refs{1};
};

std::string key_of(const char* ty, const char* id) {  // local.rs:26-29,43,61
    std::string k(ty ? ty : "");
    k += '.';
    k += id ? id : "";
    return k;
}

// service.rs:204-213: splitn(2, ":"), a record is bad when ip or port is empty
bool malformed(const std::string& a) {
    const size_t c = a.find(':');
    return c == std::string::npos || c == 0 || c + 1 >= a.size();
}

int fail(State* s, int rc, const std::string& m) {
    s->err = m;
    return rc;
}
int gp_fail(State* s, int rc) {
    s->err = rio_gp_last_error(s->gp);
    return rc;
}

int push_nodes(State* s) {
    const uint32_t m = (uint32_t)s->node_addr.size();
    int rc = rio_gp_set_nodes(s->gp, m, s->node_cap.data(), s->node_alive.data());
    return rc ? gp_fail(s, rc) : RIO_GP_OK;
}

// find or create the node id of an address; *created tells the caller to push the node table
int intern_node(State* s, const std::string& addr, bool create, uint32_t* out, bool* created) {
    auto it = s->nodes.find(addr);
    if (it != s->nodes.end()) {
        *out = it->second;
        return RIO_GP_OK;
    }
    if (!create) {
        *out = RIO_GP_NONE;
        return RIO_GP_OK;
    }
    if (s->node_addr.size() >= s->max_nodes) return fail(s, RIO_GP_EINVAL, "node table full (max_nodes)");
    const uint32_t id = (uint32_t)s->node_addr.size();
    s->nodes.emplace(addr, id);
    s->node_addr.push_back(addr);
    s->node_alive.push_back(0);  // not a member until rio_op_set_member says so (is_active == false)
    s->node_cap.push_back(RIO_GP_CAP_INF);
    const bool bad = malformed(addr);
    s->node_malformed.push_back(bad ? 1 : 0);
    s->n_malformed += bad;
    *out = id;
    if (created) *created = true;
    return RIO_GP_OK;
}

int intern_row(State* s, const char* ty, const char* id, bool create, uint32_t* out) {
    const std::string key = key_of(ty, id);
    auto it = s->rows.find(key);
    if (it != s->rows.end()) {
        *out = it->second;
        return RIO_GP_OK;
    }
    if (!create) {
        *out = RIO_GP_NONE;
        return RIO_GP_OK;
    }
    if (s->rows.size() >= s->max_objects) return fail(s, RIO_GP_EINVAL, "object table full (max_objects)");
    const uint32_t row = (uint32_t)s->rows.size();
    s->rows.emplace(key, row);
    s->row_key.emplace_back(ty ? ty : "", id ? id : "");
    *out = row;
    return RIO_GP_OK;
}

void copy_out(const std::string& v, char* out, size_t cap) {
    if (!out || !cap) return;
    const size_t n = v.size() < cap - 1 ? v.size() : cap - 1;
    memcpy(out, v.data(), n);
    out[n] = 0;
}

// the whole policy for a batch of (row, requester) pairs; rows/reqs are dense ids
int policy_batch(State* s, std::vector<uint32_t>& rows, std::vector<uint32_t>& reqs, uint32_t* out_node,
                 uint32_t* out_flag, bool tables_locked = true) {
    const uint64_t n = rows.size();
    bool any_malformed;
    {
        std::unique_lock<std::mutex> li(s->imu, std::defer_lock);
        if (!tables_locked) li.lock();
        any_malformed = s->n_malformed != 0;
    }
    if (any_malformed) {
        // service.rs:213-223: a record whose address has no ip or no port is removed (only that record)
        std::vector<uint32_t> cur(n), bad;
        int rc = rio_gp_lookup_batch(s->gp, n, rows.data(), cur.data());
        if (rc) return gp_fail(s, rc);
        {
            std::unique_lock<std::mutex> li(s->imu, std::defer_lock);
            if (!tables_locked) li.lock();
            for (uint64_t k = 0; k < n; ++k)
                if (cur[k] != RIO_GP_NONE && s->node_malformed[cur[k]]) bad.push_back(rows[k]);
        }
        if (!bad.empty() && (rc = rio_gp_remove_batch(s->gp, bad.size(), bad.data()))) return gp_fail(s, rc);
    }
    int rc = rio_gp_place_pending(s->gp, n, rows.data(), reqs.data(), out_node, out_flag);
    return rc ? gp_fail(s, rc) : RIO_GP_OK;
}

// Combining front-end for the single-object calls (ObjectPlacement::lookup, get_or_create_placement): the reference is
// called from one tokio task per connection (server.rs:292-304), and one device round trip per call (16-22 us behind a
// mutex) would cap a provider at ~5e4 calls/s however many tasks call it.  Callers queue their request; whoever finds
// nobody serving becomes the server: it takes up to kCombine queued requests, runs ONE batched device call per kind (the
// micro-batch kernels: one launch + one wait for <= 256 requests), publishes the results and repeats until the queue
// is empty.  A lone caller pays exactly what it paid before; N concurrent callers share a round trip.  Requests of one
// batch keep their arrival order (the first request for an object decides, as in rio_gp_place_pending).
constexpr size_t kCombine = 256;

void serve(State* s, std::vector<Req*>& batch) {
    std::lock_guard<std::mutex> g(s->mu);  // NOT imu: callers keep interning and queueing during the device round trip
    std::vector<uint32_t> rows, reqs, res, fl;
    std::vector<Req*> who;
    // writes first, in arrival order (sequential last-writer-wins, local.rs:22-40), then the reads and the policy calls:
    // a caller only returns after the batch, so any order inside it is a valid linearisation of concurrent calls
    for (int kind : {2, 3, 0, 1}) {
        rows.clear(); reqs.clear(); who.clear();
        for (Req* r : batch)
            if (r->kind == kind) { rows.push_back(r->row); reqs.push_back(r->req); who.push_back(r); }
        if (who.empty()) continue;
        res.assign(who.size(), RIO_GP_NONE);
        fl.assign(who.size(), 0);
        int rc;
        if (kind == 0) {
            rc = rio_gp_lookup_batch(s->gp, rows.size(), rows.data(), res.data());
            if (rc) gp_fail(s, rc);
        } else if (kind == 1) {
            rc = policy_batch(s, rows, reqs, res.data(), fl.data(), false);
        } else if (kind == 2) {
            rc = rio_gp_update_batch(s->gp, rows.size(), rows.data(), reqs.data());
            if (rc) gp_fail(s, rc);
        } else {
            rc = rio_gp_remove_batch(s->gp, rows.size(), rows.data());
            if (rc) gp_fail(s, rc);
        }
        for (size_t k = 0; k < who.size(); ++k) { who[k]->rc = rc; who[k]->node = res[k]; who[k]->flag = fl[k]; }
    }
}

int run_combined(State* s, Req* mine) {
    std::unique_lock<std::mutex> lk(s->qmu);
    s->queue.push_back(mine);
    if (s->serving) {
        // a device round trip is 15-25 us: spin on the own flag first (a futex sleep + wake costs more than the wait and,
        // with hundreds of waiters, serialises them), sleep only when it takes much longer
        lk.unlock();
        for (int spin = 0; spin < 400; ++spin) {  // ~10 us of pure spinning
            if (mine->done.load(std::memory_order_acquire)) return mine->rc;
            __builtin_ia32_pause();
        }
        for (int y = 0; y < 200; ++y) {           // then give the core away between looks (more threads than cores)
            if (mine->done.load(std::memory_order_acquire)) return mine->rc;
            sched_yield();
        }
        lk.lock();
        ++s->sleepers;
        s->qcv.wait(lk, [&] { return mine->done.load(std::memory_order_acquire) != 0; });
        --s->sleepers;
        return mine->rc;
    }
    s->serving = true;
    std::vector<Req*> batch;
    size_t last_batch = 1;
    while (!s->queue.empty()) {
        if (last_batch > 1 && s->queue.size() < last_batch) {
            // other callers are active and on their way back with their next request: a few microseconds of collecting
            // turn "one, then everyone else" into "everyone" per device round trip
            lk.unlock();
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(4)) __builtin_ia32_pause();
            lk.lock();
        }
        const size_t take = s->queue.size() < kCombine ? s->queue.size() : kCombine;
        batch.assign(s->queue.begin(), s->queue.begin() + take);
        s->queue.erase(s->queue.begin(), s->queue.begin() + take);
        lk.unlock();
        serve(s, batch);
        const int my_rc = mine->rc;  // `mine` may be in this batch: read before anything is released
        (void)my_rc;
        for (Req* r : batch)
            if (r != mine) r->done.store(1, std::memory_order_release);  // last touch of *r
        last_batch = batch.size();
        lk.lock();
        if (s->sleepers) s->qcv.notify_all();
    }
    s->serving = false;
    return mine->rc;
}

}  // namespace

struct rio_op {
    State* s;
};

extern "C" {

int rio_op_create(const rio_op_cfg* cfg, rio_op_t** out) {
    if (out) *out = nullptr;
    if (!cfg || !out || cfg->struct_size != sizeof(rio_op_cfg) || cfg->max_objects == 0 || cfg->max_nodes == 0)
        return RIO_GP_EINVAL;
    rio_gp_cfg g;
    memset(&g, 0, sizeof g);
    g.struct_size = sizeof g;
    g.device = cfg->device;
    g.max_objects = cfg->max_objects;
    g.max_nodes = cfg->max_nodes;
    g.spill_rounds = cfg->spill_rounds;
    rio_gp_t* gp = nullptr;
    int rc = rio_gp_create(&g, &gp);
    if (rc) return rc;  // text in rio_gp_last_error(NULL)
    // every potential row exists from the start, unplaced, load 1, no affinity
    if ((rc = rio_gp_set_objects(gp, cfg->max_objects, nullptr, nullptr)) || (rc = rio_gp_set_nodes(gp, 0, nullptr, nullptr))) {
        rio_gp_destroy(gp);
        return rc;
    }
    State* s = new State();
    s->gp = gp;
    s->max_objects = cfg->max_objects;
    s->max_nodes = cfg->max_nodes;
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

const char* rio_op_last_error(rio_op_t* p) { return p ? p->s->err.c_str() : rio_gp_last_error(nullptr); }

rio_gp_t* rio_op_dense(rio_op_t* p) { return p ? p->s->gp : nullptr; }

const char* rio_op_node_address(rio_op_t* p, uint32_t node_id) {
    if (!p) return nullptr;
    std::lock_guard<std::mutex> g(p->s->mu);
    std::lock_guard<std::mutex> gi(p->s->imu);
    return node_id < p->s->node_addr.size() ? p->s->node_addr[node_id].c_str() : nullptr;
}

int rio_op_update_batch(rio_op_t* p, uint64_t n, const char* const* tys, const char* const* ids,
                        const char* const* addrs) {
    if (!p || (n && (!tys || !ids || !addrs))) return RIO_GP_EINVAL;
    State* s = p->s;
    std::lock_guard<std::mutex> g(s->mu);
    std::lock_guard<std::mutex> gi(s->imu);
    std::vector<uint32_t> rows, nodes;
    rows.reserve(n);
    nodes.reserve(n);
    bool created = false;
    for (uint64_t k = 0; k < n; ++k) {
        uint32_t row = RIO_GP_NONE, node = RIO_GP_NONE;
        int rc;
        if (addrs[k]) {  // Some(address): entry(key) = address
            if ((rc = intern_row(s, tys[k], ids[k], true, &row))) return rc;
            if ((rc = intern_node(s, addrs[k], true, &node, &created))) return rc;
        } else {         // None: remove(key) (local.rs:36-37) — an unknown key stays unknown
            if ((rc = intern_row(s, tys[k], ids[k], false, &row))) return rc;
            if (row == RIO_GP_NONE) continue;
        }
        rows.push_back(row);
        nodes.push_back(node);
    }
    int rc;
    if (created && (rc = push_nodes(s))) return rc;
    if (rows.empty()) return RIO_GP_OK;
    rc = rio_gp_update_batch(s->gp, rows.size(), rows.data(), nodes.data());
    return rc ? gp_fail(s, rc) : RIO_GP_OK;
}

int rio_op_update(rio_op_t* p, const char* ty, const char* id, const char* addr) {
    if (!p) return RIO_GP_EINVAL;
    State* s = p->s;
    Req r;
    r.kind = 2;
    r.req = RIO_GP_NONE;
    bool created = false;
    {
        std::lock_guard<std::mutex> gi(s->imu);
        int rc;
        if (addr) {  // Some(address): entry(key) = address
            if ((rc = intern_row(s, ty, id, true, &r.row))) return rc;
            if ((rc = intern_node(s, addr, true, &r.req, &created))) return rc;
        } else {     // None: remove(key) (local.rs:36-37) — an unknown key stays unknown
            if ((rc = intern_row(s, ty, id, false, &r.row))) return rc;
            if (r.row == RIO_GP_NONE) return RIO_GP_OK;
        }
    }
    if (created) {  // a server address never seen before: the node table goes to the device first
        std::lock_guard<std::mutex> g(s->mu);
        std::lock_guard<std::mutex> gi(s->imu);
        int rc = push_nodes(s);
        if (rc) return rc;
    }
    return run_combined(s, &r);
}

int rio_op_lookup_batch(rio_op_t* p, uint64_t n, const char* const* tys, const char* const* ids, uint32_t* out) {
    if (!p || (n && (!tys || !ids || !out))) return RIO_GP_EINVAL;
    State* s = p->s;
    std::lock_guard<std::mutex> g(s->mu);
    std::lock_guard<std::mutex> gi(s->imu);
    std::vector<uint32_t> rows, where;
    for (uint64_t k = 0; k < n; ++k) {
        uint32_t row;
        int rc = intern_row(s, tys[k], ids[k], false, &row);
        if (rc) return rc;
        out[k] = RIO_GP_NONE;  // unknown key: Ok(None)
        if (row != RIO_GP_NONE) {
            rows.push_back(row);
            where.push_back((uint32_t)k);
        }
    }
    if (rows.empty()) return RIO_GP_OK;
    std::vector<uint32_t> res(rows.size());
    int rc = rio_gp_lookup_batch(s->gp, rows.size(), rows.data(), res.data());
    if (rc) return gp_fail(s, rc);
    for (size_t q = 0; q < rows.size(); ++q) out[where[q]] = res[q];
    return RIO_GP_OK;
}

int rio_op_lookup(rio_op_t* p, const char* ty, const char* id, char* out, size_t cap, int* found) {
    if (!p || !found) return RIO_GP_EINVAL;
    State* s = p->s;
    Req r;
    r.kind = 0;
    r.req = RIO_GP_NONE;
    {
        std::lock_guard<std::mutex> gi(s->imu);
        int rc = intern_row(s, ty, id, false, &r.row);
        if (rc) return rc;
    }
    *found = 0;
    if (r.row == RIO_GP_NONE) return RIO_GP_OK;  // unknown key: Ok(None), no device work
    int rc = run_combined(s, &r);
    if (rc) return rc;
    *found = r.node != RIO_GP_NONE;
    if (*found) {
        std::lock_guard<std::mutex> gi(s->imu);
        copy_out(s->node_addr[r.node], out, cap);
    }
    return RIO_GP_OK;
}

int rio_op_clean_server(rio_op_t* p, const char* address) {
    if (!p || !address) return RIO_GP_EINVAL;
    State* s = p->s;
    std::lock_guard<std::mutex> g(s->mu);
    std::lock_guard<std::mutex> gi(s->imu);
    uint32_t node;
    int rc = intern_node(s, address, false, &node, nullptr);
    if (rc) return rc;
    if (node == RIO_GP_NONE) return RIO_GP_OK;  // nothing was ever placed there: retain() removes nothing
    rc = rio_gp_clean_server(s->gp, node, nullptr);
    return rc ? gp_fail(s, rc) : RIO_GP_OK;
}

int rio_op_remove(rio_op_t* p, const char* ty, const char* id) {
    if (!p) return RIO_GP_EINVAL;
    State* s = p->s;
    Req r;
    r.kind = 3;
    r.req = RIO_GP_NONE;
    {
        std::lock_guard<std::mutex> gi(s->imu);
        int rc = intern_row(s, ty, id, false, &r.row);
        if (rc) return rc;
    }
    if (r.row == RIO_GP_NONE) return RIO_GP_OK;  // absent: no-op (local.rs:60-68)
    return run_combined(s, &r);
}

int rio_op_len(rio_op_t* p, uint64_t* out) {
    if (!p || !out) return RIO_GP_EINVAL;
    State* s = p->s;
    std::lock_guard<std::mutex> g(s->mu);
    std::lock_guard<std::mutex> gi(s->imu);
    int rc = rio_gp_count_placed(s->gp, out);
    return rc ? gp_fail(s, rc) : RIO_GP_OK;
}

int rio_op_set_member(rio_op_t* p, const char* address, int active, uint64_t capacity) {
    if (!p || !address) return RIO_GP_EINVAL;
    State* s = p->s;
    std::lock_guard<std::mutex> g(s->mu);
    std::lock_guard<std::mutex> gi(s->imu);
    uint32_t node;
    bool created = false;
    int rc = intern_node(s, address, true, &node, &created);
    if (rc) return rc;
    s->node_alive[node] = active ? 1 : 0;
    s->node_cap[node] = capacity;
    return push_nodes(s);
}

int rio_op_set_object_load(rio_op_t* p, const char* ty, const char* id, uint32_t load) {
    if (!p) return RIO_GP_EINVAL;
    State* s = p->s;
    std::lock_guard<std::mutex> g(s->mu);
    std::lock_guard<std::mutex> gi(s->imu);
    uint32_t row;
    int rc = intern_row(s, ty, id, true, &row);
    if (rc) return rc;
    rc = rio_gp_set_object_attrs(s->gp, 1, &row, &load, nullptr);
    return rc ? gp_fail(s, rc) : RIO_GP_OK;
}

int rio_op_get_or_create_placement_batch(rio_op_t* p, uint64_t n, const char* const* tys, const char* const* ids,
                                         const char* const* selfs, uint32_t* out_node, uint32_t* out_flag) {
    if (!p || (n && (!tys || !ids || !selfs || !out_node))) return RIO_GP_EINVAL;
    State* s = p->s;
    std::lock_guard<std::mutex> g(s->mu);
    std::lock_guard<std::mutex> gi(s->imu);
    std::vector<uint32_t> rows(n), reqs(n);
    bool created = false;
    for (uint64_t k = 0; k < n; ++k) {
        int rc;
        if ((rc = intern_row(s, tys[k], ids[k], true, &rows[k]))) return rc;
        const size_t before = s->node_addr.size();
        if ((rc = intern_node(s, selfs[k], true, &reqs[k], &created))) return rc;
        if (s->node_addr.size() != before) s->node_alive[reqs[k]] = 1;  // a server answering requests is up
    }
    int rc;
    if (created && (rc = push_nodes(s))) return rc;
    return policy_batch(s, rows, reqs, out_node, out_flag);
}

int rio_op_get_or_create_placement(rio_op_t* p, const char* ty, const char* id, const char* self_address, char* out,
                                   size_t cap, uint32_t* flag) {
    if (!p || !self_address) return RIO_GP_EINVAL;
    State* s = p->s;
    Req r;
    r.kind = 1;
    bool created = false;
    {
        std::lock_guard<std::mutex> gi(s->imu);
        int rc;
        if ((rc = intern_row(s, ty, id, true, &r.row))) return rc;
        const size_t before = s->node_addr.size();
        if ((rc = intern_node(s, self_address, true, &r.req, &created))) return rc;
        if (s->node_addr.size() != before) s->node_alive[r.req] = 1;  // a server answering requests is up
    }
    if (created) {  // rare: a requester never seen before — the node table goes to the device before the request does
        std::lock_guard<std::mutex> g(s->mu);
        std::lock_guard<std::mutex> gi(s->imu);
        int rc = push_nodes(s);
        if (rc) return rc;
    }
    int rc = run_combined(s, &r);
    if (rc) return rc;
    if (flag) *flag = r.flag;
    std::lock_guard<std::mutex> gi(s->imu);
    copy_out(r.node == RIO_GP_NONE ? std::string() : s->node_addr[r.node], out, cap);
    return RIO_GP_OK;
}

int rio_op_snapshot(rio_op_t* p, uint64_t* n_out, const char* const** struct_names, const char* const** object_ids,
                    const char* const** server_addresses) {
    if (!p || !n_out || !struct_names || !object_ids || !server_addresses) return RIO_GP_EINVAL;
    State* s = p->s;
    std::lock_guard<std::mutex> g(s->mu);
    std::lock_guard<std::mutex> gi(s->imu);
    std::vector<uint32_t> assign(s->max_objects);
    int rc = rio_gp_get_assign(s->gp, s->max_objects, assign.data());
    if (rc) return gp_fail(s, rc);
    s->snap_ty.clear(); s->snap_id.clear(); s->snap_addr.clear();
    for (size_t row = 0; row < s->row_key.size(); ++row) {
        const uint32_t nd = assign[row];
        if (nd == RIO_GP_NONE || nd >= s->node_addr.size()) continue;
        s->snap_ty.push_back(s->row_key[row].first.c_str());
        s->snap_id.push_back(s->row_key[row].second.c_str());
        s->snap_addr.push_back(s->node_addr[nd].c_str());
    }
    *n_out = s->snap_ty.size();
    *struct_names = s->snap_ty.data();
    *object_ids = s->snap_id.data();
    *server_addresses = s->snap_addr.data();
    return RIO_GP_OK;
}

int rio_op_tick(rio_op_t* p, rio_gp_stats* stats) {
    if (!p) return RIO_GP_EINVAL;
    State* s = p->s;
    std::lock_guard<std::mutex> g(s->mu);
    std::lock_guard<std::mutex> gi(s->imu);
    int rc = rio_gp_tick(s->gp, stats);
    return rc ? gp_fail(s, rc) : RIO_GP_OK;
}

}  // extern "C"

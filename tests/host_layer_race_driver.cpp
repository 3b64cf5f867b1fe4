// host_layer_race_driver.cpp — TEST INFRASTRUCTURE: hammers the string layer from many threads (single-object lookups and
// get_or_create_placement through the combining front-end, batched updates, clean_server, membership pushes, clones)
// while ThreadSanitizer watches; answers are checked against what the capacity-free policy must give.
//   phase 1  steady servers, mixed calls (the first version of this driver)
//   phase 2  EVERY call introduces or reuses a never-seen server address, from 16 threads at once: an id must never
//            reach the device ahead of its node-table entry (the stub validates like the real library)
//   phase 3  object-id churn on a small table: removed keys must be reclaimed when the table runs full, a table full
//            of LIVE objects must fail with its own error, and a failing request must not fail its batch-mates
//   phase 4  tick: removed and never-inserted rows do not come back (phantom rows)
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/rio_gpu_object_placement.h"

static rio_op_t* make(uint64_t max_objects, uint32_t max_nodes) {
    rio_op_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.max_objects = max_objects;
    cfg.max_nodes = max_nodes;
    rio_op_t* p = nullptr;
    return rio_op_create(&cfg, &p) == RIO_GP_OK ? p : nullptr;
}

int main() {
    std::atomic<int> bad{0};
    std::atomic<long> try_hits{0};
    // ---------------------------------------------------------------- phase 1
    rio_op_t* p = make(4096, 64);
    if (!p) return 1;
    for (int j = 0; j < 4; ++j) rio_op_set_member(p, ("10.0.0." + std::to_string(j) + ":5000").c_str(), 1, RIO_GP_CAP_INF);
    auto worker = [&](int tid) {
        rio_op_t* mine = rio_op_clone(p);  // every task holds its own clone (server.rs:370-392)
        char out[64];
        const std::string self = "10.0.0." + std::to_string(tid % 4) + ":5000";
        for (int k = 0; k < 400; ++k) {
            const std::string id = std::to_string((tid * 131 + k * 7) % 1000);
            uint32_t flag = 0;
            int found = 0;
            if (rio_op_get_or_create_placement(mine, "Obj", id.c_str(), self.c_str(), out, sizeof out, &flag) != RIO_GP_OK) ++bad;
            else if (!out[0] || flag > RIO_GP_FLAG_PLACED) ++bad;   // all four servers stay up: LOCAL, REDIRECT or PLACED
            if (rio_op_lookup(mine, "Obj", id.c_str(), out, sizeof out, &found) != RIO_GP_OK || !found) ++bad;
            {   // the non-blocking twins (what an async host calls inline): the shadow's answer or RIO_GP_EAGAIN, nothing else —
                // while other threads intern keys, write, clean and take snapshots
                char o2[64];
                int f2 = 0;
                uint32_t fl2 = 99;
                int rc = rio_op_try_lookup_n(mine, "Obj", 3, id.c_str(), id.size(), o2, sizeof o2, &f2);
                if (rc == RIO_GP_OK) { if (!f2 || strncmp(o2, "10.0.0.", 7) != 0) ++bad; ++try_hits; }
                else if (rc != RIO_GP_EAGAIN) ++bad;
                rc = rio_op_try_get_or_create_placement_n(mine, "Obj", 3, id.c_str(), id.size(), self.c_str(), o2, sizeof o2, &fl2);
                if (rc == RIO_GP_OK) { if (!o2[0] || fl2 > RIO_GP_FLAG_REDIRECT) ++bad; ++try_hits; }
                else if (rc != RIO_GP_EAGAIN) ++bad;
                // a key nobody has interned: Ok(None) for the lookup, EAGAIN (a first touch) for the request
                rc = rio_op_try_lookup_n(mine, "Nope", 4, id.c_str(), id.size(), o2, sizeof o2, &f2);
                if (!(rc == RIO_GP_EAGAIN || (rc == RIO_GP_OK && !f2))) ++bad;
                if (rio_op_try_get_or_create_placement_n(mine, "Nope", 4, id.c_str(), id.size(), self.c_str(), o2, sizeof o2, &fl2) != RIO_GP_EAGAIN) ++bad;
            }
            if (k % 50 == 0) {                                      // a server that was never a member: nothing to clean
                rio_op_clean_server(mine, "10.9.9.9:1");
                rio_op_set_member(mine, self.c_str(), 1, RIO_GP_CAP_INF);
            }
            if (k % 13 == 0) {  // single-object writes go through the same combining queue as the reads
                const std::string key = "w" + std::to_string(tid) + "_" + std::to_string(k);
                if (rio_op_update(mine, "Own", key.c_str(), self.c_str()) != RIO_GP_OK) ++bad;
                if (rio_op_lookup(mine, "Own", key.c_str(), out, sizeof out, &found) != RIO_GP_OK || !found || self != out) ++bad;
                if (rio_op_remove(mine, "Own", key.c_str()) != RIO_GP_OK) ++bad;
                if (rio_op_lookup(mine, "Own", key.c_str(), out, sizeof out, &found) != RIO_GP_OK || found) ++bad;
                if (rio_op_update(mine, "Own", key.c_str(), nullptr) != RIO_GP_OK) ++bad;  // None deletes (local.rs:36-37)
            }
            if (k % 97 == 0) {
                const char* ty = "Other";
                const char* oid = id.c_str();
                const char* ad = self.c_str();
                rio_op_update_batch(mine, 1, &ty, &oid, &ad);
                if (rio_op_lookup(mine, "Other", id.c_str(), out, sizeof out, &found) != RIO_GP_OK || !found) ++bad;
            }
            if (k % 101 == 0) {  // snapshots and address look-ups while the tables grow under other threads
                uint64_t n = 0;
                const char *const *ty, *const *oid, *const *ad;
                if (rio_op_snapshot(mine, &n, &ty, &oid, &ad) != RIO_GP_OK) ++bad;
                for (uint64_t q = 0; q < n; ++q)
                    if (!ty[q][0] || !ad[q][0] || strlen(oid[q]) > 40) ++bad;
                const char* a0 = rio_op_node_address(mine, 0);
                if (!a0 || strcmp(a0, "10.0.0.0:5000") != 0) ++bad;
            }
        }
        rio_op_release(mine);
    };
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 12; ++t) th.emplace_back(worker, t);
        for (auto& t : th) t.join();
    }
    uint64_t n = 0;
    rio_op_len(p, &n);
    rio_op_release(p);
    const int bad1 = bad.load();

    // ---------------------------------------------------------------- phase 2: new addresses on every call
    p = make(1 << 16, 4096);
    if (!p) return 1;
    std::atomic<int> nonok{0};
    auto newcomer = [&](int tid) {
        rio_op_t* mine = rio_op_clone(p);
        char out[64];
        for (int k = 0; k < 150; ++k) {
            // addresses are shared between threads on purpose: thread t's k-th address is thread t+1's (k-1)-th, so
            // "somebody else interned it a moment ago and has not pushed it yet" happens all the time
            const std::string a = "10.1." + std::to_string((tid + k) % 200) + "." + std::to_string(k % 7) + ":7000";
            const std::string b = "10.2." + std::to_string((tid * 3 + k) % 250) + ".1:7000";
            const std::string id = "n" + std::to_string(tid) + "_" + std::to_string(k);
            uint32_t flag = 0;
            int found = 0;
            if (rio_op_get_or_create_placement(mine, "New", id.c_str(), a.c_str(), out, sizeof out, &flag) != RIO_GP_OK) ++nonok;
            else if (a != out || flag != RIO_GP_FLAG_PLACED) ++bad;      // a brand-new object on a server that answers: first touch
            if (rio_op_update(mine, "Upd", id.c_str(), b.c_str()) != RIO_GP_OK) ++nonok;
            if (rio_op_lookup(mine, "Upd", id.c_str(), out, sizeof out, &found) != RIO_GP_OK) ++nonok;
            else if (!found || b != out) ++bad;
        }
        rio_op_release(mine);
    };
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 16; ++t) th.emplace_back(newcomer, t);
        for (auto& t : th) t.join();
    }
    rio_op_release(p);
    const int bad2 = bad.load() - bad1;

    // ---------------------------------------------------------------- phase 3: object-id churn on a small table
    p = make(512, 8);
    if (!p) return 1;
    rio_op_set_member(p, "10.3.0.1:1", 1, RIO_GP_CAP_INF);
    auto churner = [&](int tid) {
        rio_op_t* mine = rio_op_clone(p);
        char out[64];
        for (int k = 0; k < 600; ++k) {  // 8 threads x 600 distinct keys through a 512-row table: only reclamation makes it fit
            const std::string id = "c" + std::to_string(tid) + "_" + std::to_string(k);
            uint32_t flag = 0;
            int found = 1;
            if (rio_op_get_or_create_placement(mine, "Churn", id.c_str(), "10.3.0.1:1", out, sizeof out, &flag) != RIO_GP_OK) { ++nonok; continue; }
            if (strcmp(out, "10.3.0.1:1") != 0) ++bad;
            if (rio_op_remove(mine, "Churn", id.c_str()) != RIO_GP_OK) ++nonok;
            if (rio_op_lookup(mine, "Churn", id.c_str(), out, sizeof out, &found) != RIO_GP_OK || found) ++bad;
        }
        rio_op_release(mine);
    };
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 8; ++t) th.emplace_back(churner, t);
        for (auto& t : th) t.join();
    }
    // now fill the table with LIVE objects: the 513th key must fail with the table-full error, everything else keeps working
    int full_errors = 0;
    {
        char out[64];
        uint32_t flag;
        for (int k = 0; k < 520; ++k) {
            const std::string id = "live" + std::to_string(k);
            const int rc = rio_op_get_or_create_placement(p, "Live", id.c_str(), "10.3.0.1:1", out, sizeof out, &flag);
            if (rc != RIO_GP_OK) {
                ++full_errors;
                if (rc != RIO_GP_EINVAL || !strstr(rio_op_last_error(p), "object table full")) ++bad;
            }
        }
        int found = 0;
        if (rio_op_lookup(p, "Live", "live0", out, sizeof out, &found) != RIO_GP_OK || !found) ++bad;
        if (full_errors != 8) ++bad;   // 520 keys, 512 rows
        // the other entry points must report the full table too (update interns a key as well), and "nothing to do" calls
        // (unknown keys) must stay RIO_GP_OK — the two used to share a return value inside the layer
        if (rio_op_update(p, "Live", "one-too-many", "10.3.0.1:1") != RIO_GP_EINVAL) ++bad;
        if (rio_op_update(p, "Nobody", "x", nullptr) != RIO_GP_OK) ++bad;
        if (rio_op_remove(p, "Nobody", "x") != RIO_GP_OK) ++bad;
        if (rio_op_lookup(p, "Nobody", "x", out, sizeof out, &found) != RIO_GP_OK || found) ++bad;
        // a failing request next to good ones: threads looking up good keys while others run into the full table
        std::vector<std::thread> th;
        for (int t = 0; t < 6; ++t)
            th.emplace_back([&, t] {
                rio_op_t* mine = rio_op_clone(p);
                char o[64];
                for (int k = 0; k < 200; ++k) {
                    int f = 0;
                    uint32_t fl;
                    if (t % 2 == 0) {
                        if (rio_op_lookup(mine, "Live", ("live" + std::to_string(k % 500)).c_str(), o, sizeof o, &f) != RIO_GP_OK || !f) ++bad;
                    } else if (rio_op_get_or_create_placement(mine, "Over", ("o" + std::to_string(t) + "_" + std::to_string(k)).c_str(),
                                                              "10.3.0.1:1", o, sizeof o, &fl) == RIO_GP_OK) {
                        ++bad;  // the table is full of live objects: this must fail, and only this
                    }
                }
                rio_op_release(mine);
            });
        for (auto& t : th) t.join();
    }
    rio_op_release(p);
    const int bad3 = bad.load() - bad1 - bad2;

    // ---------------------------------------------------------------- phase 4: tick and rows that are not objects
    p = make(64, 4);
    if (!p) return 1;
    {
        char out[64];
        int found = 0;
        uint64_t len = 0;
        rio_gp_stats st;
        rio_op_set_member(p, "10.4.0.1:1", 1, RIO_GP_CAP_INF);
        rio_op_set_member(p, "10.4.0.2:1", 1, RIO_GP_CAP_INF);
        rio_op_update(p, "T", "a", "10.4.0.1:1");
        rio_op_update(p, "T", "b", "10.4.0.1:1");
        rio_op_update(p, "T", "c", "10.4.0.2:1");
        rio_op_remove(p, "T", "b");
        rio_op_set_object_load(p, "T", "never-placed", 5);
        if (rio_op_tick(p, &st) != RIO_GP_OK) ++bad;
        if (rio_op_lookup(p, "T", "b", out, sizeof out, &found) != RIO_GP_OK || found) ++bad;           // removed stays removed
        if (rio_op_lookup(p, "T", "never-placed", out, sizeof out, &found) != RIO_GP_OK || found) ++bad;
        if (rio_op_len(p, &len) != RIO_GP_OK || len != 2) ++bad;                                        // not max_objects
        if (st.n_objects != 2 || st.kept != 2) ++bad;
        rio_op_set_member(p, "10.4.0.1:1", 0, RIO_GP_CAP_INF);                                          // a's server dies
        if (rio_op_tick(p, &st) != RIO_GP_OK || st.evicted != 1) ++bad;
        if (rio_op_lookup(p, "T", "a", out, sizeof out, &found) != RIO_GP_OK || !found || strcmp(out, "10.4.0.2:1") != 0) ++bad;
        if (rio_op_len(p, &len) != RIO_GP_OK || len != 2) ++bad;
    }
    rio_op_release(p);
    const int bad4 = bad.load() - bad1 - bad2 - bad3;

    if (try_hits.load() == 0) ++bad;  // (the shadow answered none of thousands of repeated lookups: the try calls are dead)
    printf("wrong=%d (phase1 %d phase2 %d phase3 %d phase4 %d) nonok=%d placed=%llu try_hits=%ld\n", bad.load(), bad1, bad2, bad3, bad4,
           nonok.load(), (unsigned long long)n, try_hits.load());
    return (bad.load() || nonok.load()) ? 2 : 0;
}

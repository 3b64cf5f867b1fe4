// host_layer_race_driver.cpp — TEST INFRASTRUCTURE: hammers the string layer from many threads (single-object lookups and
// get_or_create_placement through the combining front-end, batched updates, clean_server, membership pushes, clones)
// while ThreadSanitizer watches; answers are checked against what the capacity-free policy must give.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/rio_gpu_object_placement.h"

int main() {
    rio_op_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.max_objects = 4096;
    cfg.max_nodes = 64;
    rio_op_t* p = nullptr;
    if (rio_op_create(&cfg, &p) != RIO_GP_OK) return 1;
    for (int j = 0; j < 4; ++j) rio_op_set_member(p, ("10.0.0." + std::to_string(j) + ":5000").c_str(), 1, RIO_GP_CAP_INF);
    std::atomic<int> bad{0};
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
        }
        rio_op_release(mine);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < 12; ++t) th.emplace_back(worker, t);
    for (auto& t : th) t.join();
    uint64_t n = 0;
    rio_op_len(p, &n);
    rio_op_release(p);
    printf("wrong=%d placed=%llu\n", bad.load(), (unsigned long long)n);
    return bad.load() ? 2 : 0;
}

/*
 * local_placement_oracle.cpp — string-level CPU restatement of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may link or call this.  The product never does.
 *
 * The reference is Rust and there is no Rust toolchain here, so this is a line-for-line
 * C++ restatement (kind = "port"), pinned against the reference's own known-answer tests
 * in tests/test_oracle_golden.py:
 *
 *   LocalObjectPlacement            /root/reference/rio-rs/src/object_placement/local.rs:12-68
 *     HashMap<String,String> behind RwLock, key = format!("{}.{}", type, id) (local.rs:26-29)
 *   MembershipStorage::is_active    /root/reference/rio-rs/src/cluster/storage/mod.rs:95-110
 *     fetch ALL members, retain active, linear scan comparing ip and port strings
 *   Service::get_or_create_placement   /root/reference/rio-rs/src/service.rs:193-254
 *   Service::check_address_mismatch    /root/reference/rio-rs/src/service.rs:261-298
 */
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

// local.rs:12  type PlacementMap = Arc<RwLock<HashMap<String, String>>>
struct PlacementMap {
    std::shared_mutex lock;
    std::unordered_map<std::string, std::string> map;
};

// local.rs:15-18  #[derive(Default, Clone, Debug)] struct LocalObjectPlacement { placement }
struct LocalObjectPlacement {
    std::shared_ptr<PlacementMap> placement = std::make_shared<PlacementMap>();

    static std::string key(const char* ty, const char* id) {  // local.rs:26-29,43,61
        std::string k(ty);
        k += '.';
        k += id;
        return k;
    }
    // local.rs:22-40
    void update(const char* ty, const char* id, const char* address) {
        std::string object_id = key(ty, id);
        std::unique_lock<std::shared_mutex> g(placement->lock);
        if (address) placement->map[object_id] = address;  // *entry(object_id).or_default() = address
        else placement->map.erase(object_id);              // None => remove (local.rs:36-37)
    }
    // local.rs:42-49
    bool lookup(const char* ty, const char* id, std::string* out) const {
        std::string object_id = key(ty, id);
        std::shared_lock<std::shared_mutex> g(placement->lock);
        auto it = placement->map.find(object_id);
        if (it == placement->map.end()) return false;
        *out = it->second;  // .cloned()
        return true;
    }
    // local.rs:51-58  retain(|_, v| *v != address): full scan under the write lock
    void clean_server(const std::string& address) {
        std::unique_lock<std::shared_mutex> g(placement->lock);
        for (auto it = placement->map.begin(); it != placement->map.end();) {
            if (it->second == address) it = placement->map.erase(it);
            else ++it;
        }
    }
    // local.rs:60-68
    void remove(const char* ty, const char* id) {
        std::string object_id = key(ty, id);
        std::unique_lock<std::shared_mutex> g(placement->lock);
        placement->map.erase(object_id);
    }
};

// cluster/storage/mod.rs:20-58  Member { ip, port, active, .. }; address() = "{ip}:{port}"
struct Member {
    std::string ip, port;
    bool active;
};
// cluster/storage/local.rs:10-17  LocalStorage { members: Arc<RwLock<Vec<Member>>> }
struct LocalStorage {
    std::shared_mutex lock;
    std::vector<Member> members;

    // mod.rs:95-99 active_members(): members().await (a CLONE of the Vec, local.rs:61-63) + retain(active)
    std::vector<Member> active_members() {
        std::vector<Member> v;
        {
            std::shared_lock<std::shared_mutex> g(lock);
            v = members;
        }
        std::vector<Member> out;
        for (auto& x : v)
            if (x.active) out.push_back(std::move(x));
        return out;
    }
    // mod.rs:102-110
    bool is_active(const std::string& ip, const std::string& port) {
        for (const auto& mbr : active_members())
            if (mbr.ip == ip && mbr.port == port) return true;
        return false;
    }
    void set_is_active(const std::string& ip, const std::string& port, bool active) {
        std::unique_lock<std::shared_mutex> g(lock);
        for (auto& x : members)
            if (x.ip == ip && x.port == port) x.active = active;
    }
};

// service.rs:193-254
std::string get_or_create_placement(LocalObjectPlacement& provider, LocalStorage& members_storage,
                                    const std::string& self_address, const char* handler_type,
                                    const char* handler_id) {
    std::string server_address;
    bool some = provider.lookup(handler_type, handler_id, &server_address);  // :199-201
    if (some) {
        // :204-207  splitn(2, ":")
        size_t colon = server_address.find(':');
        std::string ip = colon == std::string::npos ? server_address : server_address.substr(0, colon);
        std::string port = colon == std::string::npos ? std::string() : server_address.substr(colon + 1);
        if (ip.empty() || port.empty()) {  // :213-223 bad record -> remove
            provider.remove(handler_type, handler_id);
            some = false;
        } else if (!members_storage.is_active(ip, port)) {  // :227-237 -> clean_server
            provider.clean_server(server_address);
            some = false;
        }
    }
    if (some) return server_address;  // :241-242 sticky
    provider.update(handler_type, handler_id, self_address.c_str());  // :244-252 first touch
    return self_address;
}

// service.rs:261-298 -> 0 Ok, 1 Redirect(addr), 2 DeallocateServiceObject, -1 malformed
int check_address_mismatch(LocalObjectPlacement& provider, LocalStorage& members_storage,
                           const std::string& self_address, const std::string& server_address) {
    if (server_address == self_address) return 0;  // :262-264
    // :266-278  split(':') -> ip = first piece, port = second piece
    size_t c1 = server_address.find(':');
    if (c1 == std::string::npos) return -1;  // Missing PORT
    std::string ip = server_address.substr(0, c1);
    size_t c2 = server_address.find(':', c1 + 1);
    std::string port = server_address.substr(c1 + 1, c2 == std::string::npos ? std::string::npos : c2 - c1 - 1);
    if (members_storage.is_active(ip, port)) return 1;  // :280-289
    provider.clean_server(server_address);              // :292-296
    return 2;
}

std::string node_address(uint32_t j) {  // SURVEY.md §8d synthetic node address
    return "10." + std::to_string(j >> 16) + "." + std::to_string((j >> 8) & 255) + "." +
           std::to_string(j & 255) + ":5000";
}
uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

}  // namespace

extern "C" {

void* lpo_new() { return new LocalObjectPlacement(); }
void* lpo_clone(void* p) { return new LocalObjectPlacement(*static_cast<LocalObjectPlacement*>(p)); }  // Clone shares the Arc
void lpo_free(void* p) { delete static_cast<LocalObjectPlacement*>(p); }
int lpo_update(void* p, const char* ty, const char* id, const char* addr) {
    static_cast<LocalObjectPlacement*>(p)->update(ty, id, addr);
    return 0;
}
// returns 1 = Some (address copied into buf), 0 = None
int lpo_lookup(void* p, const char* ty, const char* id, char* buf, size_t buflen) {
    std::string out;
    if (!static_cast<LocalObjectPlacement*>(p)->lookup(ty, id, &out)) return 0;
    if (buf && buflen) {
        std::strncpy(buf, out.c_str(), buflen - 1);
        buf[buflen - 1] = 0;
    }
    return 1;
}
int lpo_clean_server(void* p, const char* addr) {
    static_cast<LocalObjectPlacement*>(p)->clean_server(addr);
    return 0;
}
int lpo_remove(void* p, const char* ty, const char* id) {
    static_cast<LocalObjectPlacement*>(p)->remove(ty, id);
    return 0;
}
uint64_t lpo_len(void* p) {
    auto* o = static_cast<LocalObjectPlacement*>(p);
    std::shared_lock<std::shared_mutex> g(o->placement->lock);
    return o->placement->map.size();
}

void* mem_new() { return new LocalStorage(); }
void mem_free(void* p) { delete static_cast<LocalStorage*>(p); }
void mem_push(void* p, const char* ip, const char* port, int active) {
    auto* s = static_cast<LocalStorage*>(p);
    std::unique_lock<std::shared_mutex> g(s->lock);
    s->members.push_back(Member{ip, port, active != 0});
}
void mem_set_is_active(void* p, const char* ip, const char* port, int active) {
    static_cast<LocalStorage*>(p)->set_is_active(ip, port, active != 0);
}
int mem_is_active(void* p, const char* ip, const char* port) {
    return static_cast<LocalStorage*>(p)->is_active(ip, port) ? 1 : 0;
}

int lpo_get_or_create_placement(void* p, void* mem, const char* self_addr, const char* ty, const char* id,
                                char* buf, size_t buflen) {
    std::string r = get_or_create_placement(*static_cast<LocalObjectPlacement*>(p),
                                            *static_cast<LocalStorage*>(mem), self_addr, ty, id);
    if (buf && buflen) {
        std::strncpy(buf, r.c_str(), buflen - 1);
        buf[buflen - 1] = 0;
    }
    return 0;
}
int lpo_check_address_mismatch(void* p, void* mem, const char* self_addr, const char* server_addr) {
    return check_address_mismatch(*static_cast<LocalObjectPlacement*>(p), *static_cast<LocalStorage*>(mem),
                                  self_addr, server_addr);
}

/*
 * cpu_baseline leg: the reference's per-object path — one get_or_create_placement per object
 * (ObjectId("Obj", i), requester = node aff[i]) against ONE shared LocalObjectPlacement and
 * ONE shared LocalStorage of m active members, from `threads` threads (mirrors many tokio
 * tasks on one Arc<RwLock<P>>, server.rs:103-104).  Cold start: every call misses, first
 * touches, updates.  Returns seconds; *decisions = number of calls made.
 * If warm != 0 a second, timed pass repeats the same requests (all sticky hits) instead.
 */
double lpo_bench_policy(uint64_t n_objects, uint32_t m_nodes, const uint32_t* aff, int threads, int warm,
                        uint64_t* decisions) {
    LocalObjectPlacement provider;
    LocalStorage storage;
    std::vector<std::string> addr(m_nodes);
    for (uint32_t j = 0; j < m_nodes; ++j) {
        addr[j] = node_address(j);
        size_t c = addr[j].find(':');
        storage.members.push_back(Member{addr[j].substr(0, c), addr[j].substr(c + 1), true});
    }
    if (threads < 1) threads = 1;
    auto run = [&](bool timed) {
        std::vector<std::thread> th;
        auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < threads; ++t) {
            th.emplace_back([&, t]() {
                uint64_t lo = n_objects * t / threads, hi = n_objects * (t + 1) / threads;
                for (uint64_t i = lo; i < hi; ++i) {
                    std::string id = std::to_string(i);
                    std::string r = get_or_create_placement(provider, storage, addr[aff[i] % m_nodes], "Obj", id.c_str());
                    if (r.empty()) std::abort();
                }
            });
        }
        for (auto& x : th) x.join();
        auto t1 = std::chrono::steady_clock::now();
        (void)timed;
        return std::chrono::duration<double>(t1 - t0).count();
    };
    double s = run(true);
    if (warm) s = run(true);
    if (decisions) *decisions = n_objects;
    return s;
}

/*
 * The same per-object path with its RESULT kept (bench.py's parity.against_reference_port, tests/): one shared
 * LocalObjectPlacement, optionally pre-populated from cur[] (a warm table: update("Obj", i -> address of cur[i]) for
 * every cur[i] != NONE, local.rs:22-40), one LocalStorage of m members whose `active` flags come from alive[] (nullptr:
 * all active); then ONE get_or_create_placement per object, in index order, requester = node aff[i] (service.rs:193-254:
 * an object found on an inactive server has that server cleaned — every entry of it, local.rs:51-58 — and is first-touched
 * on the requester); then every object is looked up again (local.rs:42-49) and its address turned back into the node's
 * index: out[i] = node, or 0xFFFFFFFF for a miss.  Returns the seconds the get_or_create_placement calls took
 * (single thread).
 */
double lpo_policy_readback(uint64_t n_objects, uint32_t m_nodes, const uint32_t* aff, const uint8_t* alive,
                           const uint32_t* cur, uint32_t* out) {
    LocalObjectPlacement provider;
    LocalStorage storage;
    std::vector<std::string> addr(m_nodes);
    std::unordered_map<std::string, uint32_t> index_of;
    for (uint32_t j = 0; j < m_nodes; ++j) {
        addr[j] = node_address(j);
        index_of[addr[j]] = j;
        size_t c = addr[j].find(':');
        storage.members.push_back(Member{addr[j].substr(0, c), addr[j].substr(c + 1), alive ? alive[j] != 0 : true});
    }
    if (cur)
        for (uint64_t i = 0; i < n_objects; ++i)
            if (cur[i] < m_nodes) provider.update("Obj", std::to_string(i).c_str(), addr[cur[i]].c_str());
    auto t0 = std::chrono::steady_clock::now();
    for (uint64_t i = 0; i < n_objects; ++i) {
        std::string id = std::to_string(i);
        std::string r = get_or_create_placement(provider, storage, addr[aff[i] % m_nodes], "Obj", id.c_str());
        if (r.empty()) std::abort();
    }
    auto t1 = std::chrono::steady_clock::now();
    for (uint64_t i = 0; i < n_objects; ++i) {
        std::string a;
        if (!provider.lookup("Obj", std::to_string(i).c_str(), &a)) { out[i] = 0xFFFFFFFFu; continue; }
        auto it = index_of.find(a);
        out[i] = it == index_of.end() ? 0xFFFFFFFEu : it->second;
    }
    return std::chrono::duration<double>(t1 - t0).count();
}

/* clean_server on a populated map: n_objects entries spread over m_nodes addresses; times
 * retain() for one address (local.rs:51-58).  Returns seconds. */
double lpo_bench_clean_server(uint64_t n_objects, uint32_t m_nodes, const uint32_t* node_of, uint32_t victim) {
    LocalObjectPlacement provider;
    std::vector<std::string> addr(m_nodes);
    for (uint32_t j = 0; j < m_nodes; ++j) addr[j] = node_address(j);
    for (uint64_t i = 0; i < n_objects; ++i) {
        std::string id = std::to_string(i);
        provider.update("Obj", id.c_str(), addr[node_of[i] % m_nodes].c_str());
    }
    auto t0 = std::chrono::steady_clock::now();
    provider.clean_server(addr[victim]);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

uint64_t lpo_splitmix64(uint64_t x) { return splitmix64(x); }
unsigned lpo_hardware_concurrency() { return std::thread::hardware_concurrency(); }

}  // extern "C"

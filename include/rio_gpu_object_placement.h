/*
 * rio_gpu_object_placement.h — string layer of the C ABI: the ObjectPlacement trait itself.
 *
 * What a Rust `GpuObjectPlacement: ObjectPlacement` binds (rio-rs_amd/rust/).
 * Keys and values are the reference's own types (paths relative to /root/reference):
 *
 *   ObjectId(struct_name, object_id)        rio-rs/src/service_object.rs:19-26
 *   ObjectPlacementItem{.., Option<String>} rio-rs/src/object_placement/mod.rs:20-34
 *   trait ObjectPlacement                   rio-rs/src/object_placement/mod.rs:38-56
 *   LocalObjectPlacement (parity target)    rio-rs/src/object_placement/local.rs:22-68
 *   Service::get_or_create_placement        rio-rs/src/service.rs:193-254
 *
 * Strings are interned to dense rows / node ids on the host and every call lands in the
 * rio_gp_* layer (rio_gpu_placement.h); the assignment lives in HBM only.  The object key is
 * "{struct_name}.{object_id}" exactly as local.rs:26-29 builds it — including its quirk that
 * ("a.b","c") and ("a","b.c") are the same object.
 *
 * Same conventions as rio_gpu_placement.h: int rc (RIO_GP_OK / EINVAL -> Unknown / EUPSTREAM ->
 * Upstream), nothing thrown across the ABI, handle internally synchronized, no CPU fallback.
 */
#ifndef RIO_GPU_OBJECT_PLACEMENT_H
#define RIO_GPU_OBJECT_PLACEMENT_H

#include "rio_gpu_placement.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rio_op rio_op_t;

typedef struct rio_op_cfg {
    uint32_t struct_size;  /* = sizeof(rio_op_cfg) */
    int32_t device;
    uint64_t max_objects;  /* distinct object keys the table can hold */
    uint32_t max_nodes;    /* distinct server addresses */
    uint32_t spill_rounds; /* 0 -> 2 */
    uint32_t flags;        /* 0 = the reference's behaviour: get_or_create_placement first-touches the requester whether or not
                            * membership marks it active, as service.rs:244-252 does (the dense layer's
                            * RIO_GP_CFG_REF_SELF_ASSIGN; that bit is accepted here and changes nothing).
                            * RIO_OP_CFG_LIVE_FIRST_TOUCH opts OUT: a requester that membership marks inactive is not a
                            * placement target, its first touches go to the water-fill (the capacity-aware extension) */
    uint32_t collect_ns;   /* single-object calls that need the device share round trips (flat combining): how long, at most, a
                            * thread that takes over the device waits until about as many callers have published as the last two
                            * batches carried together, in ns; 0 = the default (RIO_OP_DEFAULT_COLLECT_NS), 1 = do not wait;
                            * more than RIO_OP_MAX_COLLECT_NS is RIO_GP_EINVAL (a field a version-1 client left uninitialised) */
} rio_op_cfg;
#define RIO_OP_DEFAULT_COLLECT_NS 6000u
#define RIO_OP_MAX_COLLECT_NS 1000000u
#define RIO_OP_CFG_LIVE_FIRST_TOUCH 4u
/* Host shadow of the assignment column.  The reference calls lookup / get_or_create_placement once per request from one task per
 * connection (server.rs:292-304) and LocalObjectPlacement answers a hit from a hash map (local.rs:42-49); a device round trip per
 * call is 8-11 us however well concurrent callers share it.  So the string layer remembers, per row, the last answer the DEVICE
 * gave (node + a validity stamp) and answers a later rio_op_lookup of that key — and a rio_op_get_or_create_placement whose object
 * sits on a server that is an active member: the sticky path of service.rs:199-242 — from there.  It never decides anything:
 * first touches, evictions, capacities and ticks are the GPU's, and whatever may move rows the layer cannot name (clean_server,
 * rio_op_tick, a request that ran into a dead server, reclaimed keys) invalidates by stamp.  This flag switches the shadow off
 * (every call goes to the device): A/B measurements, examples/c_host_threads.c. */
#define RIO_OP_CFG_NO_HOST_SHADOW 8u

/* LocalObjectPlacement::default() + ObjectPlacement::prepare (local.rs:15-18, mod.rs:42-44). */
int rio_op_create(const rio_op_cfg* cfg, rio_op_t** out);
/* #[derive(Clone)]: the clone SHARES the map (Arc inside, local.rs:12-18; pinned by
 * local.rs:71-123).  Returns the same state with one more reference. */
rio_op_t* rio_op_clone(rio_op_t* p);
/* Drop: releases one reference; the HBM tables go with the last one. */
void rio_op_release(rio_op_t* p);
/* mod.rs:42-44 (default no-op; kept so the adapter can forward it). */
int rio_op_prepare(rio_op_t* p);
const char* rio_op_last_error(rio_op_t* p);

/* ObjectPlacement::update (mod.rs:46-49, local.rs:22-40): upsert; server_address == NULL is
 * Option::None and deletes the entry (local.rs:36-37). */
int rio_op_update(rio_op_t* p, const char* struct_name, const char* object_id, const char* server_address);
/* ObjectPlacement::lookup (mod.rs:50, local.rs:42-49): *found = 1 and the address copied into
 * out (NUL-terminated) or *found = 0 (Ok(None)).  The reference returns an owned String of any
 * length, so an address is NEVER truncated: when it does not fit out_cap the call returns
 * RIO_GP_ERANGE with *found = 1 and out = "", and rio_op_last_address_len says how long the
 * address is — the caller asks again with a buffer of that length + 1 (lookup is a pure read). */
int rio_op_lookup(rio_op_t* p, const char* struct_name, const char* object_id, char* out, size_t out_cap,
                  int* found);
/* Length in bytes (without the NUL) of the address the CALLING THREAD's last rio_op_lookup /
 * rio_op_get_or_create_placement produced (0: Ok(None) / UNPLACED). */
size_t rio_op_last_address_len(rio_op_t* p);
/* ObjectPlacement::clean_server (mod.rs:52, local.rs:51-58). */
int rio_op_clean_server(rio_op_t* p, const char* address);
/* ObjectPlacement::remove (mod.rs:55, local.rs:60-68). */
int rio_op_remove(rio_op_t* p, const char* struct_name, const char* object_id);
/* number of placed objects (HashMap::len of local.rs:12) */
int rio_op_len(rio_op_t* p, uint64_t* out);

/* Batched forms of the same calls: n keys at once, one kernel launch per call. */
int rio_op_update_batch(rio_op_t* p, uint64_t n, const char* const* struct_names, const char* const* object_ids,
                        const char* const* server_addresses);
/* out_node_ids[k] = node id or RIO_GP_NONE; resolve ids with rio_op_node_address. */
int rio_op_lookup_batch(rio_op_t* p, uint64_t n, const char* const* struct_names, const char* const* object_ids,
                        uint32_t* out_node_ids);
const char* rio_op_node_address(rio_op_t* p, uint32_t node_id);

/* Membership feed: MembershipStorage::push / set_is_active (cluster/storage/mod.rs:74-80) pushed
 * into the node table instead of being polled per request (is_active, mod.rs:102-110).  An address
 * only ever seen through `update` is not a member, i.e. not active, as in the reference.
 * capacity: load units, RIO_GP_CAP_INF = unbounded (the reference has no capacity). */
int rio_op_set_member(rio_op_t* p, const char* address, int active, uint64_t capacity);
/* per-object load (default 1; new behaviour, the reference has none).  A key first seen here keeps its row (and the
 * load) until its first update / request makes it an object; from then on it is reclaimed like any other key. */
int rio_op_set_object_load(rio_op_t* p, const char* struct_name, const char* object_id, uint32_t load);

/* Service::get_or_create_placement (service.rs:193-254) + check_address_mismatch (service.rs:261-298)
 * for one request arriving at server `self_address`: returns the address the object lives on now and
 * *flag = RIO_GP_FLAG_{LOCAL,REDIRECT,PLACED,SPILLED,UNPLACED} (UNPLACED: out is ""), with RIO_GP_FLAG_REPLACED OR-ed on
 * when the object was found on a server that is not alive (that server was cleaned, the object re-placed).
 * RIO_GP_ERANGE (address longer than out_cap - 1): the decision IS made and *flag is set, out = ""; fetch the
 * address with rio_op_lookup into a buffer of rio_op_last_address_len() + 1 bytes. */
int rio_op_get_or_create_placement(rio_op_t* p, const char* struct_name, const char* object_id,
                                   const char* self_address, char* out, size_t out_cap, uint32_t* flag);
/* The same for n requests at once, processed as if sequentially in array order. */
int rio_op_get_or_create_placement_batch(rio_op_t* p, uint64_t n, const char* const* struct_names,
                                         const char* const* object_ids, const char* const* self_addresses,
                                         uint32_t* out_node_ids, uint32_t* out_flags);

/* Durable twin of the table: every placed entry as (struct_name, object_id, server_address) —
 * the columns of the reference's object_placement table (migrations/0001-sqlite-init.sql:1-9; written by
 * sqlite.rs:68-85).  The arrays belong to the handle and stay valid until its next call.  Loading a snapshot is
 * rio_op_update_batch.  rio-rs_amd/snapshot.py moves them to and from a SQLite file in that schema, so a GPU-backed
 * server can warm-start from, or write back to, the placement DB of a SqliteObjectPlacement deployment. */
int rio_op_snapshot(rio_op_t* p, uint64_t* n_out, const char* const** struct_names, const char* const** object_ids,
                    const char* const** server_addresses);

/* Keys with their lengths.  ObjectId(String, String) (service_object.rs:19-26) holds any Rust string, a NUL byte included;
 * the entry points above take NUL-terminated strings and would cut such a key short.  These take struct_name / object_id
 * as (pointer, length) and are otherwise the same calls (the Rust adapter binds THESE: rio-rs_amd/rust/src/gpu.rs).
 * Server addresses stay NUL-terminated: an address is "{ip}:{port}" of a Member (cluster/storage/mod.rs:56-58).
 * rio_op_snapshot hands out NUL-terminated copies; rio_op_snapshot_key_lengths gives the true lengths of the struct_name /
 * object_id strings of the CALLING THREAD's last snapshot (arrays of n entries, valid until its next snapshot). */
int rio_op_update_n(rio_op_t* p, const char* struct_name, size_t struct_name_len, const char* object_id, size_t object_id_len,
                    const char* server_address);
int rio_op_lookup_n(rio_op_t* p, const char* struct_name, size_t struct_name_len, const char* object_id, size_t object_id_len,
                    char* out, size_t out_cap, int* found);
int rio_op_remove_n(rio_op_t* p, const char* struct_name, size_t struct_name_len, const char* object_id, size_t object_id_len);
int rio_op_get_or_create_placement_n(rio_op_t* p, const char* struct_name, size_t struct_name_len, const char* object_id,
                                     size_t object_id_len, const char* self_address, char* out, size_t out_cap, uint32_t* flag);
/* The two calls an unchanged Server makes for EVERY request (server.rs:292-304 -> service.rs:199-200: lookup, then the sticky
 * branch of get_or_create_placement), answered from the host shadow ONLY: they never touch the device, never wait for a lock a
 * device call can hold, never intern anything.  RIO_GP_OK: answered, exactly as rio_op_lookup_n / rio_op_get_or_create_placement_n
 * would have (a key nobody has interned is Ok(None) for the lookup); RIO_GP_EAGAIN: the shadow cannot say — an unknown key or
 * requester for the request, a row whose last answer has been invalidated, an object that is pending or sits on a server that is
 * not an active, well-formed member, the table lock held by a writer — nothing was done, make the blocking call.  An async host
 * calls these inline on its worker thread and pays the hand-off to a blocking thread (tokio::task::spawn_blocking: several
 * microseconds) only on EAGAIN; LocalObjectPlacement::lookup never yields either (local.rs:42-49).  With
 * RIO_OP_CFG_NO_HOST_SHADOW every call is EAGAIN. */
int rio_op_try_lookup_n(rio_op_t* p, const char* struct_name, size_t struct_name_len, const char* object_id, size_t object_id_len,
                        char* out, size_t out_cap, int* found);
int rio_op_try_get_or_create_placement_n(rio_op_t* p, const char* struct_name, size_t struct_name_len, const char* object_id,
                                         size_t object_id_len, const char* self_address, char* out, size_t out_cap, uint32_t* flag);
int rio_op_snapshot_key_lengths(rio_op_t* p, const size_t** struct_name_lens, const size_t** object_id_lens);
/* ... and the batched calls: struct_names[k] / object_ids[k] point at struct_name_lens[k] / object_id_lens[k] bytes (any byte, a
 * NUL included); otherwise rio_op_update_batch / rio_op_lookup_batch / rio_op_get_or_create_placement_batch /
 * rio_op_set_object_load.  Loading a snapshot whose keys may hold NUL bytes is rio_op_update_batch_n. */
int rio_op_update_batch_n(rio_op_t* p, uint64_t n, const char* const* struct_names, const size_t* struct_name_lens,
                          const char* const* object_ids, const size_t* object_id_lens, const char* const* server_addresses);
int rio_op_lookup_batch_n(rio_op_t* p, uint64_t n, const char* const* struct_names, const size_t* struct_name_lens,
                          const char* const* object_ids, const size_t* object_id_lens, uint32_t* out_node_ids);
int rio_op_get_or_create_placement_batch_n(rio_op_t* p, uint64_t n, const char* const* struct_names, const size_t* struct_name_lens,
                                           const char* const* object_ids, const size_t* object_id_lens,
                                           const char* const* self_addresses, uint32_t* out_node_ids, uint32_t* out_flags);
int rio_op_set_object_load_n(rio_op_t* p, const char* struct_name, size_t struct_name_len, const char* object_id, size_t object_id_len,
                             uint32_t load);

/* Whole-table re-solve over the interned tables (rio_gp_tick): the eager form of the reference's lazy clean_server + first-touch
 * path — every object that sits on a server that is not an active member is evicted and re-placed at once.  A tick has no
 * requester that vouches for itself, so it places on active members only, whatever rio_op_cfg.flags says about requests. */
int rio_op_tick(rio_op_t* p, rio_gp_stats* stats);
/* The dense handle underneath (borrowed).  A mutation made through it bypasses the host shadow: follow it with
 * rio_op_invalidate_cache. */
rio_gp_t* rio_op_dense(rio_op_t* p);
/* Drop everything the host shadow holds (RIO_OP_CFG_NO_HOST_SHADOW's comment): the next lookup of every key asks the device. */
int rio_op_invalidate_cache(rio_op_t* p);
/* How often the single-object calls went to the device so far: combined batches (one round trip each) and the requests they
 * carried.  calls made - *requests = calls answered from the host shadow or without any device work. */
int rio_op_device_round_trips(rio_op_t* p, uint64_t* batches, uint64_t* requests);

#ifdef __cplusplus
}
#endif
#endif /* RIO_GPU_OBJECT_PLACEMENT_H */

/*
 * rio_gpu_placement.h — C ABI of the MI355X batched object-placement solver.
 *
 * This is the drop-in boundary for rio-rs' `object_placement::*` hot path.  Nothing like it
 * exists in the reference (pure Rust, static dispatch); these are the entry points an FFI
 * binding for that path would call.  Every entry point cites the reference interface it
 * replaces (paths relative to /root/reference):
 *
 *   trait ObjectPlacement            rio-rs/src/object_placement/mod.rs:38-56
 *   LocalObjectPlacement (semantics) rio-rs/src/object_placement/local.rs:22-68
 *   placement policy                 rio-rs/src/service.rs:193-298
 *   error convention                 rio-rs/src/errors.rs:135-142
 *
 * Two layers are exported from the same shared library (librio_gp.so):
 *
 *   rio_gp_*  dense-index layer.  Objects are rows 0..n-1, nodes are 0..m-1, the table
 *             (object: cur/load/aff) x (node: cap/alive/used) lives in HBM.  Plain
 *             uint32_t/uint64_t arrays, caller-owned, copied in/out.  `_dev` variants take
 *             device pointers (inputs already resident in HBM).
 *   rio_op_*  string layer (rio_gpu_object_placement.h): (struct_name, object_id) and
 *             "ip:port" strings, i.e. exactly the ObjectPlacement trait; interns strings
 *             to dense ids on the host and calls the rio_gp_* layer.
 *
 * Conventions
 *   - Every call returns int: 0 = ok.  Non-zero HIP status -> RIO_GP_EUPSTREAM
 *     (ObjectPlacementError::Upstream); bad argument/index -> RIO_GP_EINVAL
 *     (ObjectPlacementError::Unknown).  A lookup miss is RIO_GP_NONE in the output with
 *     rc 0, never an error (local.rs:48).  Nothing throws or aborts across the ABI.
 *   - A handle is internally synchronized (one mutex + one HIP stream per handle); calls
 *     are synchronous on return unless the name ends in `_async`.
 *     "Synchronous" means the results are in the caller's buffers and every effect is
 *     ordered before the handle's next call; the library waits on completion words its
 *     kernels store into mapped pinned memory (the calling thread spins for the length of
 *     the call — call from a blocking pool, not from an async executor thread) and asks
 *     the HIP stream only when a word does not arrive (a device fault surfaces there).
 *   - There is NO CPU fallback: rio_gp_create fails with RIO_GP_ENODEV without a gfx950
 *     device.
 */
#ifndef RIO_GPU_PLACEMENT_H
#define RIO_GPU_PLACEMENT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: rio_op_cfg.flags == 0 means the reference's first touch (round 5 flipped the default: RIO_OP_CFG_LIVE_FIRST_TOUCH opts out),
 * rio_op_cfg.reserved became collect_ns, rio_gp_mixed_batch and the rio_op_*_batch_n / rio_op_try_* calls exist, RIO_GP_EAGAIN.
 * A client built against version 1 that passes flags = 0 gets another placement policy: compare rio_gp_abi_version() with the
 * header it was built against. */
#define RIO_GP_ABI_VERSION 2u

/* "not placed": Option::None of lookup (object_placement/mod.rs:50). */
#define RIO_GP_NONE 0xFFFFFFFFu
/* hard limits of the solver */
#define RIO_GP_MAX_NODES 8192u
#define RIO_GP_MAX_OBJECTS 0x7FFFF000ull
/* capacity value meaning "unbounded" (the reference has no capacity at all) */
#define RIO_GP_CAP_INF 0xFFFFFFFFFFFFFFFFull
/* Affinity value of a row that is NOT an object: the dense twin of "no entry in the map" (a key that was never
 * inserted, or was removed: local.rs:36-37,60-68, or dropped by clean_server: local.rs:51-58).  A whole-table solve
 * keeps such a row where it is if it happens to be placed on a live node and never places it otherwise; it is not
 * counted in rio_gp_stats.n_objects.  Any other affinity >= the node count (RIO_GP_NONE included) means "an object
 * without a preferred node": it goes to the water-fill. */
#define RIO_GP_AFF_INACTIVE 0xFFFFFFFEu
/* rio_gp_cfg.flags */
/* Row lifecycle (what the string layer uses): rows start as non-objects (affinity RIO_GP_AFF_INACTIVE) and the CRUD
 * calls maintain that column — update(i, node) and the first place_pending request of a pending row make row i an
 * object (affinity = the node / the requester), update(i, NONE), remove and clean_server make it a non-object again.
 * Without the flag the affinity column is only ever written by rio_gp_set_objects / rio_gp_set_object_attrs. */
#define RIO_GP_CFG_ROW_LIFECYCLE 1u
/* Reference-faithful first touch.  By default a node (a requester) that membership marks inactive is not a placement target:
 * a pending object whose affinity node / requester is not alive goes to the water-fill.  The reference has no such test —
 * Service::get_or_create_placement updates the object onto self.address whatever the membership says about self
 * (service.rs:244-252), and evicts it again on the next touch (service.rs:227-237).  With this flag the solver does the
 * same: a pending row claims its affinity node (a place_pending request: its requester) whether or not that node is alive,
 * against the node's whole capacity; rows on nodes that are not alive are still evicted, and the water-fill still places
 * on live nodes only.  With unbounded capacities rio_gp_tick / rio_gp_place_pending then equal the reference request by
 * request for ANY membership (tests/test_gpu_object_placement.py).  Single-GPU handles only: the rio_gp_shard_* calls
 * return RIO_GP_EINVAL on a handle created with it. */
#define RIO_GP_CFG_REF_SELF_ASSIGN 2u

/* return codes */
#define RIO_GP_OK 0
#define RIO_GP_EINVAL 1    /* -> ObjectPlacementError::Unknown  (errors.rs:140-141) */
#define RIO_GP_EUPSTREAM 2 /* -> ObjectPlacementError::Upstream (errors.rs:137-138) */
#define RIO_GP_ENODEV 3    /* no HIP device / not gfx950: the product path fails loudly */
#define RIO_GP_ENOMEM 4
#define RIO_GP_ERANGE 5    /* string layer: the caller's output buffer is too small (nothing is truncated) */
#define RIO_GP_EAGAIN 6    /* string layer, rio_op_try_*: the host shadow cannot answer without the device (or without a lock a device
                            * call may hold): nothing was done, make the blocking call */

/* per-request outcome of rio_gp_place_pending (service.rs:193-298 folded into one code) */
#define RIO_GP_FLAG_LOCAL 0u      /* already placed on the requester (sticky hit, service.rs:241-242,262-264) */
#define RIO_GP_FLAG_REDIRECT 1u   /* already placed on another live node (ResponseError::Redirect, service.rs:286-289) */
#define RIO_GP_FLAG_PLACED 2u     /* was unplaced/evicted, now first-touch placed on the requester (service.rs:244-252) */
#define RIO_GP_FLAG_SPILLED 3u    /* requester full: placed on another node (capacity extension; caller redirects) */
#define RIO_GP_FLAG_UNPLACED 4u   /* no capacity anywhere: stays RIO_GP_NONE */
/* OR-ed onto PLACED / SPILLED / UNPLACED of the request that found its object on a server that is not alive: that server
 * was cleaned (clean_server, service.rs:227-237) and the object re-placed by this call — check_address_mismatch's
 * "placed elsewhere, but there is dead" branch (service.rs:268-285).  Later requests for the same object in the same batch
 * observe the new placement (LOCAL / REDIRECT). */
#define RIO_GP_FLAG_REPLACED 0x10u
#define RIO_GP_FLAG_MASK 0x0Fu    /* the five outcomes above */

typedef struct rio_gp rio_gp_t;

typedef struct rio_gp_cfg {
    uint32_t struct_size;  /* = sizeof(rio_gp_cfg); ABI guard */
    int32_t device;        /* HIP device ordinal */
    uint64_t max_objects;  /* row capacity of the object table (<= RIO_GP_MAX_OBJECTS) */
    uint32_t max_nodes;    /* row capacity of the node table (<= RIO_GP_MAX_NODES) */
    uint32_t spill_rounds; /* water-fill rounds for objects their affinity node rejects; 0 -> default 2 */
    uint32_t flags;        /* RIO_GP_CFG_* bits, 0 = none */
    uint32_t reserved;
} rio_gp_cfg;

/* Counters of one whole-table solve (rio_gp_tick / rio_gp_solve). */
typedef struct rio_gp_stats {
    uint64_t n_objects; /* rows that are objects = kept + claimed + spilled + unplaced */
    uint64_t kept;      /* sticky: placed on a live node (service.rs:241-242) */
    uint64_t evicted;   /* were placed on a dead node (clean_server, service.rs:227-237) */
    uint64_t claimed;   /* pending, admitted on their affinity node (first touch, service.rs:244-252) */
    uint64_t spilled;   /* pending, placed on another node by the water-fill */
    uint64_t unplaced;  /* pending, no room: stay RIO_GP_NONE */
    uint64_t load_kept, load_claimed, load_spilled, load_unplaced;
    uint32_t cut_nodes;   /* nodes whose claimants exceeded their free capacity */
    uint32_t slow_path;   /* 0 = single-pass fast path, 1 = cut/spill fix-up ran */
    uint32_t rounds_run;  /* spill rounds executed */
    uint32_t reserved;
} rio_gp_stats;

/* ---- lifetime -------------------------------------------------------------------------- */

/* ObjectPlacement::prepare (mod.rs:42-44; SQL migrations sqlite.rs:58-66): allocate the HBM
 * tables, create the stream.  `*out` is NULL on failure; rio_gp_last_error(NULL) has the text. */
int rio_gp_create(const rio_gp_cfg* cfg, rio_gp_t** out);
void rio_gp_destroy(rio_gp_t* h);
/* Text of the last failure on this handle (or of the last failed create when h == NULL).
 * The Rust adapter wraps it as ObjectPlacementError::{Upstream,Unknown}(text). */
const char* rio_gp_last_error(rio_gp_t* h);
/* Change RIO_GP_CFG_REF_SELF_ASSIGN after creation (the only bit that may change; every other bit of `flags` must equal the
 * handle's).  The string layer uses it around rio_op_tick: a whole-table solve has no requester that vouches for itself, so it
 * places on active members only even when requests self-assign.  Counts as a change of the solve's inputs. */
int rio_gp_set_flags(rio_gp_t* h, uint32_t flags);
/* "hip:gfx950" — there is no other backend. */
const char* rio_gp_backend(rio_gp_t* h);
uint32_t rio_gp_abi_version(void);
/* hipStreamSynchronize on the handle's stream. */
int rio_gp_sync(rio_gp_t* h);

/* ---- node table: the "node" side, fed by MembershipStorage (cluster/storage/mod.rs:70-121) */

/* Replace the node table: m nodes, capacity (load units) and liveness (Member.active,
 * cluster/storage/mod.rs:20-58).  cap == NULL -> all RIO_GP_CAP_INF; alive == NULL -> all 1. */
int rio_gp_set_nodes(rio_gp_t* h, uint32_t m, const uint64_t* cap, const uint8_t* alive);
/* MembershipStorage::set_is_active (cluster/storage/mod.rs:80) pushed instead of polled
 * (is_active, mod.rs:102-110).  A push costs no device work of its own: the bitmap waits in mapped pinned memory and
 * reaches the device with the next whole-table solve or tick (whose scan reads it from there), or with one tiny kernel
 * when a request batch needs it first.  Every later call sees the new liveness. */
int rio_gp_set_alive(rio_gp_t* h, uint32_t node, uint8_t alive);
int rio_gp_set_alive_all(rio_gp_t* h, uint32_t m, const uint8_t* alive);
int rio_gp_get_nodes(rio_gp_t* h, uint32_t m, uint64_t* cap, uint8_t* alive, uint64_t* used);

/* ---- object table ---------------------------------------------------------------------- */

/* Define rows 0..n-1: per-object load and affinity (= the requesting server `self.address`
 * of service.rs:244).  All rows start unplaced.  load == NULL -> 1; aff == NULL -> RIO_GP_NONE
 * (RIO_GP_AFF_INACTIVE under RIO_GP_CFG_ROW_LIFECYCLE). */
int rio_gp_set_objects(rio_gp_t* h, uint64_t n, const uint32_t* load, const uint32_t* aff);
int rio_gp_set_objects_dev(rio_gp_t* h, uint64_t n, const uint32_t* d_load, const uint32_t* d_aff);
/* Bulk-load / dump the whole assignment column (warm start, snapshot; the on-disk twin is
 * object_placement(struct_name, object_id, server_address), migrations/0001-sqlite-init.sql:1-9). */
int rio_gp_set_assign(rio_gp_t* h, uint64_t n, const uint32_t* assign);
int rio_gp_set_assign_dev(rio_gp_t* h, uint64_t n, const uint32_t* d_assign);
int rio_gp_get_assign(rio_gp_t* h, uint64_t n, uint32_t* out_assign);
/* Read the load and/or affinity columns back (either may be NULL). */
int rio_gp_get_objects(rio_gp_t* h, uint64_t n, uint32_t* out_load, uint32_t* out_aff);
/* Change load and/or affinity of individual rows (either array may be NULL = leave as is). */
int rio_gp_set_object_attrs(rio_gp_t* h, uint64_t n, const uint32_t* idx, const uint32_t* load,
                            const uint32_t* aff);
/* Change how many rows take part (0 <= n <= max_objects) WITHOUT touching their contents: rows >= n are neither
 * solved nor valid indices until n grows again.  A host that hands out rows one by one (the string layer's interning)
 * keeps n at its high-water mark so that a solve streams the rows in use, not the table's capacity. */
int rio_gp_set_num_objects(rio_gp_t* h, uint64_t n);
/* Number of placed rows (HashMap::len of local.rs:12): one 4 B/row pass. */
int rio_gp_count_placed(rio_gp_t* h, uint64_t* out);
/* Device pointer of the live assignment column (valid until the next tick/commit). */
const uint32_t* rio_gp_assign_dev(rio_gp_t* h);
uint64_t rio_gp_num_objects(rio_gp_t* h);
uint32_t rio_gp_num_nodes(rio_gp_t* h);

/* ---- the ObjectPlacement CRUD, batched -------------------------------------------------- */

/* ObjectPlacement::lookup (mod.rs:50; local.rs:42-49): out_node[k] = node of idx[k] or
 * RIO_GP_NONE.  idx[k] >= n is RIO_GP_EINVAL. */
int rio_gp_lookup_batch(rio_gp_t* h, uint64_t n, const uint32_t* idx, uint32_t* out_node);
int rio_gp_lookup_batch_dev(rio_gp_t* h, uint64_t n, const uint32_t* d_idx, uint32_t* d_out_node);
/* ObjectPlacement::update (mod.rs:46-49; local.rs:22-40): upsert idx[k] -> node[k];
 * node[k] == RIO_GP_NONE deletes (local.rs:36-37).  Duplicate idx in one batch resolve as the
 * sequential loop would: the highest batch position wins. */
int rio_gp_update_batch(rio_gp_t* h, uint64_t n, const uint32_t* idx, const uint32_t* node);
int rio_gp_update_batch_dev(rio_gp_t* h, uint64_t n, const uint32_t* d_idx, const uint32_t* d_node);
/* ObjectPlacement::remove (mod.rs:55; local.rs:60-68): un-place; absent is a no-op. */
int rio_gp_remove_batch(rio_gp_t* h, uint64_t n, const uint32_t* idx);
int rio_gp_remove_batch_dev(rio_gp_t* h, uint64_t n, const uint32_t* d_idx);
/* ObjectPlacement::clean_server (mod.rs:52; local.rs:51-58): un-place EVERY object whose
 * node == `node` (one coalesced pass over the assignment column). */
int rio_gp_clean_server(rio_gp_t* h, uint32_t node, uint64_t* evicted);
/* Same for any number of failed nodes in ONE pass: bit j of dead_bitmap (ceil(m/64) words). */
int rio_gp_clean_servers(rio_gp_t* h, const uint64_t* dead_bitmap, uint64_t* evicted);

/* ---- the placement policy, batched ------------------------------------------------------ */

/* Service::get_or_create_placement + check_address_mismatch (service.rs:193-298) for a batch
 * of requests, processed as if sequentially in batch order: request k asks for object idx[k]
 * on server requester[k].
 *   - object placed on a node that is not alive -> clean_server(that node) (all of its objects
 *     are un-placed, service.rs:233-237), the object becomes pending;
 *   - still placed -> sticky: out_node = that node, flag LOCAL / REDIRECT;
 *   - pending -> first touch on requester[k] (service.rs:244-252), admitted while the
 *     requester's free capacity lasts (index-ordered prefix rule), else
 *     water-filled onto the emptiest nodes, else left RIO_GP_NONE.
 * With every capacity = RIO_GP_CAP_INF this is exactly the reference policy.  out_flag may
 * be NULL. */
int rio_gp_place_pending(rio_gp_t* h, uint64_t n, const uint32_t* idx, const uint32_t* requester,
                         uint32_t* out_node, uint32_t* out_flag);
/* The same with request and result arrays already resident in HBM (device pointers; d_out_flag may be NULL).  The
 * entries are checked on the device BEFORE anything is changed: an out-of-range index or requester fails the whole call
 * with RIO_GP_EINVAL and leaves the table as it was. */
int rio_gp_place_pending_dev(rio_gp_t* h, uint64_t n, const uint32_t* d_idx, const uint32_t* d_requester,
                             uint32_t* d_out_node, uint32_t* d_out_flag);

/* Up to 256 entries of EACH of update / remove / lookup / place_pending in one call: exactly rio_gp_update_batch, then
 * rio_gp_remove_batch, then rio_gp_lookup_batch, then rio_gp_place_pending over the given arrays (a count of 0 skips the
 * kind) — but ONE launch and ONE host wait for all of them (one workgroup runs the parts one after the other).
 * This is what the connections of one reference Server ask for at the same moment (lookups, first touches and removals
 * mixed: service.rs:193-254, server.rs:292-304; local.rs:22-68); the string layer's combiner sends one such call per
 * generation of concurrent callers.  rc[k] = what the k-th of those calls would have returned (0 update, 1 remove, 2 lookup,
 * 3 place_pending): a kind with an out-of-range entry changes nothing and reports RIO_GP_EINVAL there, the other kinds still
 * run.  The return value is about the call as a whole (arguments, device). */
typedef struct rio_gp_mixed {
    uint32_t struct_size; /* sizeof(rio_gp_mixed) */
    uint32_t n_update;
    const uint32_t* update_idx;
    const uint32_t* update_node;
    uint32_t n_remove;
    uint32_t n_lookup;
    const uint32_t* remove_idx;
    const uint32_t* lookup_idx;
    uint32_t* lookup_out;
    uint32_t n_place;
    uint32_t reserved;
    const uint32_t* place_idx;
    const uint32_t* place_requester;
    uint32_t* place_node;
    uint32_t* place_flag; /* may be NULL */
    int32_t rc[4];
} rio_gp_mixed;
int rio_gp_mixed_batch(rio_gp_t* h, rio_gp_mixed* ops);

/* Whole-table solve: every row gets a decision in one call (the eager form of the lazy
 * per-request path of service.rs:193-254 + the clean_server stream of SURVEY §3.2):
 * keep if alive (sticky) | claim affinity node by index-ordered prefix | water-fill | NONE.
 *   rio_gp_solve   computes the new assignment column and `used`, does not publish it;
 *   rio_gp_commit  publishes the last solve (pointer swap);
 *   rio_gp_tick    = solve + commit. */
int rio_gp_solve(rio_gp_t* h, rio_gp_stats* stats);
int rio_gp_commit(rio_gp_t* h);
int rio_gp_tick(rio_gp_t* h, rio_gp_stats* stats);
/* The committed tick without a host wait: rio_gp_tick_async enqueues solve + fix-up + commit of one tick on the handle's
 * stream and returns; ticks enqueued back to back each consume the previous one's commit on the device (and any
 * rio_gp_set_alive* pushed in between, in stream order).  rio_gp_tick_wait waits for all of them and hands out their
 * counters, oldest first: *n_out = ticks completed since the last wait, out[0..min(cap, *n_out)) = the most recent ones.
 * The tables are exactly what the same sequence of rio_gp_tick calls produces; only the counters arrive later.  (A
 * server that pushes a membership change and rebalances has no use for the counters before the next push.)
 * On the handle's OWN stream a tick that cannot need the fix-up (nothing has changed since a tick that left every object
 * placed) runs its resolve step beside the next tick's scan, and on tables of 2^18 rows and more (up to 1 024 nodes) the
 * scans of such ticks alternate between the handle's stream and a second internal one (the resolve steps in line behind their
 * scans below 5 M rows, on a third stream behind the scan's completion from there on), each
 * wave starting as soon as the same wave of the previous tick's scan has finished its rows (a dependency per row
 * range instead of per launch: the data dependency between two ticks is exactly that).  Every other call of the handle
 * orders itself behind all of it.  On a caller's stream (rio_gp_set_stream) everything stays on that one stream. */
int rio_gp_tick_async(rio_gp_t* h);
int rio_gp_tick_wait(rio_gp_t* h, rio_gp_stats* out, uint32_t cap, uint32_t* n_out);
/* Enqueue one solve on the handle's stream without waiting.  rio_gp_solve_wait drains the
 * stream, runs the cut/spill fix-up for the LAST enqueued solve if it needed one, and
 * returns its stats.  *n_slow = how many of the solves enqueued since the last wait took
 * the fix-up path on the device. */
int rio_gp_solve_async(rio_gp_t* h);
int rio_gp_solve_wait(rio_gp_t* h, rio_gp_stats* stats, uint32_t* n_slow);
/* Device pointer of the column the last solve produced (before commit). */
const uint32_t* rio_gp_solved_dev(rio_gp_t* h);
int rio_gp_get_solved(rio_gp_t* h, uint64_t n, uint32_t* out_assign);

/* ---- row-sharded solve across the GPUs of one node ----------------------- */
/*
 * Rank r (one process, one GPU, one handle) owns the contiguous rows [off_r, off_r + n_r) of the
 * object table; shard order = index order; the node table is replicated.  Rows couple only through
 * the per-node load vectors, so the only cross-rank data are two small records, all-gathered by
 * the CALLER between the calls below (RCCL over xGMI from the Rust/Python host; the library itself
 * never talks to another rank):
 *   X (rio_gp_shard_words1 = 2m+8 u64): local kept load[m] | local claim load[m] | 8 counters
 *   Y (rio_gp_shard_words2 =  m+2 u64): locally admitted load[m] | pending spill load | pending rows
 * Every cross-rank reduction is an integer sum taken in rank order, so the sharded solve equals the
 * unsharded rio_gp_solve of the concatenated table bit for bit.
 *
 *   fast path (nothing cut, nothing spills):   scan -> [all-gather X] -> resolve -> verdict -> finish
 *   fix-up:   ... verdict -> cut -> [all-gather Y] -> merge
 *                 -> { spill(round) -> [all-gather Y] -> merge } while rows are pending -> finish
 * then rio_gp_commit publishes as usual.  d_* are DEVICE pointers owned by the caller.
 */
typedef struct rio_gp_shard_info {
    uint64_t cut_nodes;    /* nodes whose global claim load exceeds their free capacity */
    uint64_t spill_rows;   /* rows (all ranks) whose affinity node is unusable: go to the water-fill */
    uint64_t local_fixup;  /* != 0: THIS rank has claimants to re-mark (pass it to rio_gp_shard_cut) */
    uint64_t kept, evicted, claimants, load_kept, load_claim; /* global row/load counters */
} rio_gp_shard_info;
/* Enqueue every later call of this handle on the caller's HIP stream (hipStream_t as void*; NULL = back
 * to the handle's own stream), e.g. the stream the host's RCCL all-gather is ordered against. */
int rio_gp_set_stream(rio_gp_t* h, void* hip_stream);
uint32_t rio_gp_shard_words1(rio_gp_t* h);
uint32_t rio_gp_shard_words2(rio_gp_t* h);
/* k_scan + local column sums; d_x[words1] = this rank's X record.  Asynchronous. */
int rio_gp_shard_scan(rio_gp_t* h, uint64_t* d_x);
/* d_xg[n_ranks][words1] = the all-gathered X records.  Asynchronous.  on_stream (hipStream_t, may be NULL = the
 * handle's stream): a host that pipelines independent solves runs the exchange and this step on a second stream,
 * ordered after rio_gp_shard_scan by its own event, so the all-gather overlaps the next solve's scan. */
int rio_gp_shard_resolve(rio_gp_t* h, uint32_t rank, uint32_t n_ranks, const uint64_t* d_xg, void* on_stream);
/* Waits for the stream; global verdict of the LAST resolve; *n_slow = how many of the resolves enqueued
 * since the last verdict need the fix-up. */
int rio_gp_shard_verdict(rio_gp_t* h, rio_gp_shard_info* out, uint32_t* n_slow);
/* Fix-up step 1 (only if cut_nodes or spill_rows): exact cut on this rank, d_y[words2] = Y record. */
int rio_gp_shard_cut(rio_gp_t* h, int run_local_fixup, uint64_t* d_y);
/* d_yg[n_ranks][words2] = all-gathered Y records -> global `used`, this rank's spill base.  Synchronous;
 * returns the rows / load still waiting for the water-fill on all ranks. */
int rio_gp_shard_merge(rio_gp_t* h, const uint64_t* d_yg, uint64_t* pending_rows, uint64_t* pending_load);
/* One water-fill round over this rank's pending rows; d_y[words2] = Y record of the round. */
int rio_gp_shard_spill(rio_gp_t* h, uint32_t round, int last, uint64_t* d_y);
/* Local counters of this rank's rows (sum them over ranks; cut_nodes/slow_path/rounds_run are global
 * quantities the caller already holds).  After this, rio_gp_commit / rio_gp_get_solved work as usual. */
int rio_gp_shard_finish(rio_gp_t* h, rio_gp_stats* local_stats);

/* Preferred on one node: peer-to-peer exchange over xGMI with no collective call at all.  Every rank exports an
 * uncached window (rio_gp_shard_p2p_export -> 64-byte hipIpcMemHandle_t), the host all-gathers the handles over its
 * control channel, every rank maps its peers' windows (rio_gp_shard_p2p_connect; ends with a handshake and fails
 * with RIO_GP_EUPSTREAM if a peer's store does not become visible within 3 s — fall back to RCCL then).  After that
 * rio_gp_shard_solve_async is two launches on one stream, no host involvement: the scan, then one kernel in which every
 * workgroup stores its eight nodes' sums straight into the peers' HBM as data-tagged 8-byte words, polls the same words of
 * every rank and resolves its nodes (no flag, no collective).  rio_gp_shard_exchange (fix-up records) stores the raw
 * record and a sequence flag; the consuming kernel waits on the flags.  All ranks must call export/connect/solve/exchange
 * collectively and in the same order. */
int rio_gp_shard_p2p_export(rio_gp_t* h, uint32_t n_ranks, void* out_handle64);
int rio_gp_shard_p2p_connect(rio_gp_t* h, uint32_t rank, uint32_t n_ranks, const void* handles /* [n_ranks][64] */);
int rio_gp_shard_p2p_ready(rio_gp_t* h);
/* Unmap the peers' windows and free ours (e.g. to fall back to RCCL after a time-out). */
int rio_gp_shard_p2p_close(rio_gp_t* h);

/* One COMMITTED tick of the row-sharded table with nothing waiting on the host (peer-to-peer windows only; the sharded twin
 * of rio_gp_tick_async): the scan and the one-launch exchange, then the whole fix-up chain — exact cut on this rank, Y
 * record out, everyone's in, one water-fill round per spill round, each followed by its exchange — every step guarded on the
 * device, so a tick that needs none of it pays a handful of short launches, and the publication (two pointer swaps).  The
 * result is what the synchronous sequence above computes.  All ranks must enqueue the same ticks in the same order; at most
 * 64 may be in flight.  rio_gp_shard_tick_wait waits for them and returns the records of the last `cap` (oldest first),
 * *n_out = how many had been enqueued: this rank's counters (sum them over the ranks) and the global verdict of each tick. */
typedef struct rio_gp_shard_tick_info {
    rio_gp_stats local;   /* this rank's rows (cut_nodes / slow_path / rounds_run of it: the global values below) */
    uint64_t cut_nodes;   /* global */
    uint64_t spill_rows;  /* global: rows that went to the water-fill before any cut */
    uint32_t slow_path;   /* the tick needed the fix-up (on any rank) */
    uint32_t rounds_run;  /* water-fill rounds that had rows pending */
} rio_gp_shard_tick_info;
int rio_gp_shard_tick_async(rio_gp_t* h);
int rio_gp_shard_tick_wait(rio_gp_t* h, rio_gp_shard_tick_info* out, uint32_t cap, uint32_t* n_out);

/* Optional: let the library issue the all-gathers itself through RCCL (resolved at run time with dlopen — the
 * copy already in the process, else librccl.so.1; `rccl_path` may name one, NULL = default search).  The host only
 * moves the 128-byte ncclUniqueId from rank 0 to the other ranks over its own control channel:
 *   rank 0: rio_gp_shard_comm_unique_id(id);  all ranks: rio_gp_shard_comm_init(h, rank, n_ranks, id).
 * rio_gp_shard_solve_async = scan -> all-gather X -> resolve in ONE call, the exchange on a second stream so that
 * back-to-back independent solves overlap it with the next scan; follow with rio_gp_shard_verdict / _finish (or the
 * fix-up calls, using rio_gp_shard_exchange for the Y records) exactly as above. */
int rio_gp_shard_comm_unique_id(void* out128, const char* rccl_path);
int rio_gp_shard_comm_init(rio_gp_t* h, uint32_t rank, uint32_t n_ranks, const void* id128, const char* rccl_path);
uint32_t rio_gp_shard_comm_ranks(rio_gp_t* h);
int rio_gp_shard_solve_async(rio_gp_t* h);
int rio_gp_shard_exchange(rio_gp_t* h, const uint64_t* d_in, uint64_t* d_out, uint64_t words_per_rank);

/* ---- measurement hooks (HIP events on the handle's own stream) -------------------------- */
int rio_gp_timer_begin(rio_gp_t* h);
/* rio_gp_timer_stop records the closing event behind the work enqueued so far and returns at once; rio_gp_timer_end waits
 * for the closing event (recording it first when nobody has) and reports the milliseconds between the two. */
int rio_gp_timer_stop(rio_gp_t* h);
int rio_gp_timer_end(rio_gp_t* h, float* ms);
/* One fast-path solve with a HIP-event pair around EACH kernel launch (so the durations are
 * per-launch, not host-paired): scan_ms = the streaming kernel k_scan (16 algorithmic B/row),
 * resolve_ms = k_resolve.  Does not publish; fails with RIO_GP_EINVAL if the solve needs the
 * cut/spill fix-up (use rio_gp_solve then). */
int rio_gp_solve_profiled(rio_gp_t* h, float* scan_ms, float* resolve_ms);
/* A/B knobs, phase traces and the streaming probes are not part of the boundary: rio_gpu_placement_debug.h. */

#ifdef __cplusplus
}
#endif
#endif /* RIO_GPU_PLACEMENT_H */

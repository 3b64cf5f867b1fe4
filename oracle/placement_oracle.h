/*
 * placement_oracle.h — CPU oracle of the batched object-placement solver (dense-index layer).
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may link or call this; the product (librio_gp.so) never does.
 *
 * Parity status
 *   - Map semantics (lookup/update/remove/clean_server) restate
 *     /root/reference/rio-rs/src/object_placement/local.rs:22-68 over dense ids and are
 *     pinned, through oracle/local_placement_oracle.cpp, against the reference's own
 *     known-answer tests (tests/test_oracle_golden.py).
 *   - The policy with every capacity = infinity restates
 *     /root/reference/rio-rs/src/service.rs:193-254 and is pinned against the string-level
 *     restatement of that function on identical request streams.
 *   - Everything about load / capacity / affinity / spill ordering is NEW behaviour the
 *     reference does not have: **parity unpinned** — this file is its definition
 *     (DESIGN.md §Spec), the HIP kernels must match it bit for bit.
 */
#ifndef RIO_PLACEMENT_ORACLE_H
#define RIO_PLACEMENT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NONE 0xFFFFFFFFu
#define ORC_CAP_INF 0xFFFFFFFFFFFFFFFFull
/* affinity value of a row that is not an object (RIO_GP_AFF_INACTIVE): a solve keeps it where it is if it is
 * placed on a live node, and never places it otherwise; it is not counted in n_objects */
#define ORC_AFF_INACTIVE 0xFFFFFFFEu
#define ORC_FLAG_REPLACED 0x10u  /* OR-ed onto the outcome of a request that found its object on a dead server (service.rs:268-285) */

/* identical layout to rio_gp_stats (include/rio_gpu_placement.h) */
typedef struct orc_stats {
    uint64_t n_objects, kept, evicted, claimed, spilled, unplaced;
    uint64_t load_kept, load_claimed, load_spilled, load_unplaced;
    uint32_t cut_nodes, slow_path, rounds_run, reserved;
} orc_stats;

/* local.rs:42-49 */
int orc_lookup_batch(const uint32_t* assign, uint64_t n_obj, const uint32_t* idx, uint64_t n,
                     uint32_t* out_node);
/* local.rs:22-40 (sequential loop: last writer wins; node NONE deletes) */
int orc_update_batch(uint32_t* assign, uint64_t n_obj, uint32_t m, const uint32_t* idx,
                     const uint32_t* node, uint64_t n);
/* local.rs:60-68 */
int orc_remove_batch(uint32_t* assign, uint64_t n_obj, const uint32_t* idx, uint64_t n);
/* local.rs:51-58 for every node whose bit is set; returns number of evicted rows */
uint64_t orc_clean_servers(uint32_t* assign, uint64_t n_obj, const uint64_t* dead_bitmap, uint32_t m);
/* used[j] = sum of load over rows assigned to j */
void orc_recompute_used(const uint32_t* assign, const uint32_t* load, uint64_t n_obj, uint32_t m,
                        uint64_t* used);

/* Whole-table solve (service.rs:193-254 generalised with capacity). */
int orc_tick(const uint32_t* cur, const uint32_t* load, const uint32_t* aff, uint64_t n_obj,
             const uint64_t* cap, const uint8_t* alive, uint32_t m, uint32_t rounds,
             uint32_t* next, uint64_t* used_out, orc_stats* st);

/* flags of the _ex forms: ORC_REF_SELF_ASSIGN = RIO_GP_CFG_REF_SELF_ASSIGN — a claim / first touch does not need an active
 * node (service.rs:244-252 self-assigns without asking is_active(self)) */
#define ORC_REF_SELF_ASSIGN 2u
int orc_tick_ex(const uint32_t* cur, const uint32_t* load, const uint32_t* aff, uint64_t n_obj,
                const uint64_t* cap, const uint8_t* alive, uint32_t m, uint32_t rounds, uint32_t flags,
                uint32_t* next, uint64_t* used_out, orc_stats* st);
int orc_place_pending_ex(uint32_t* assign, const uint32_t* load, uint64_t n_obj, const uint64_t* cap,
                         const uint8_t* alive, uint64_t* used, uint32_t m, uint32_t rounds, uint32_t flags,
                         const uint32_t* idx, const uint32_t* requester, uint64_t n,
                         uint32_t* out_node, uint32_t* out_flag);

/* Batched get_or_create_placement (service.rs:193-298), see rio_gp_place_pending. */
int orc_place_pending(uint32_t* assign, const uint32_t* load, uint64_t n_obj, const uint64_t* cap,
                      const uint8_t* alive, uint64_t* used, uint32_t m, uint32_t rounds,
                      const uint32_t* idx, const uint32_t* requester, uint64_t n,
                      uint32_t* out_node, uint32_t* out_flag);

/* Deterministic synthetic inputs: r(i,k) = splitmix64(seed ^ (k<<56) ^ i). */
uint64_t orc_splitmix64(uint64_t x);

#ifdef __cplusplus
}
#endif
#endif

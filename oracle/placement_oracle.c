/*
 * placement_oracle.c — sequential CPU definition of the batched placement solver.
 *
 * TEST INFRASTRUCTURE ONLY (see placement_oracle.h).  Plain C, no threads, no SIMD: every
 * pass is the obvious loop so that it can be read against the reference:
 *
 *   map semantics   /root/reference/rio-rs/src/object_placement/local.rs:22-68
 *   policy          /root/reference/rio-rs/src/service.rs:193-298
 *   liveness        /root/reference/rio-rs/src/cluster/storage/mod.rs:95-110
 *
 * Capacity / load / spill are new behaviour: **parity unpinned**, defined here.
 */
#include "placement_oracle.h"

#include <stdlib.h>
#include <string.h>

uint64_t orc_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* ---- map semantics over dense ids ------------------------------------------------------- */

/* local.rs:42-49: map.get(key).cloned(); a miss is None, not an error. */
int orc_lookup_batch(const uint32_t* assign, uint64_t n_obj, const uint32_t* idx, uint64_t n,
                     uint32_t* out_node) {
    for (uint64_t k = 0; k < n; ++k)
        if (idx[k] >= n_obj) return 1;
    for (uint64_t k = 0; k < n; ++k) out_node[k] = assign[idx[k]];
    return 0;
}

/* local.rs:22-40: Some(addr) -> entry(key) = addr (upsert), None -> remove(key).
 * Applied in batch order, so the last writer of a key wins. */
int orc_update_batch(uint32_t* assign, uint64_t n_obj, uint32_t m, const uint32_t* idx,
                     const uint32_t* node, uint64_t n) {
    for (uint64_t k = 0; k < n; ++k)
        if (idx[k] >= n_obj || (node[k] != ORC_NONE && node[k] >= m)) return 1;
    for (uint64_t k = 0; k < n; ++k) assign[idx[k]] = node[k];
    return 0;
}

/* local.rs:60-68: remove(key); absent is a no-op. */
int orc_remove_batch(uint32_t* assign, uint64_t n_obj, const uint32_t* idx, uint64_t n) {
    for (uint64_t k = 0; k < n; ++k)
        if (idx[k] >= n_obj) return 1;
    for (uint64_t k = 0; k < n; ++k) assign[idx[k]] = ORC_NONE;
    return 0;
}

/* local.rs:51-58: retain(|_, v| *v != address), for a set of addresses at once. */
uint64_t orc_clean_servers(uint32_t* assign, uint64_t n_obj, const uint64_t* dead_bitmap, uint32_t m) {
    uint64_t evicted = 0;
    for (uint64_t i = 0; i < n_obj; ++i) {
        uint32_t c = assign[i];
        if (c != ORC_NONE && c < m && ((dead_bitmap[c >> 6] >> (c & 63)) & 1ull)) {
            assign[i] = ORC_NONE;
            ++evicted;
        }
    }
    return evicted;
}

void orc_recompute_used(const uint32_t* assign, const uint32_t* load, uint64_t n_obj, uint32_t m,
                        uint64_t* used) {
    memset(used, 0, (size_t)m * sizeof(uint64_t));
    for (uint64_t i = 0; i < n_obj; ++i) {
        uint32_t c = assign[i];
        if (c != ORC_NONE && c < m) used[c] += load[i];
    }
}

/* ---- water-fill (spill) ------------------------------------------------------------------ */

/* Order of the nodes in the water-fill: by CAPACITY CLASS descending, node index ascending inside a class.  The class of a
 * free capacity f > 0 is f rounded down to three significant bits, as an ordinal: 4 * floor(log2 f) + the two bits below
 * the leading one (256 classes, monotone in f).  Emptiest nodes first, to within a quarter octave — and a rank that a
 * workgroup obtains by COUNTING (histogram over classes + position among the equals) instead of a comparison sort of m
 * 64-bit keys, which costs ~1 400 vector instructions per key on one CU (measured: 10 us at m = 1 024).  DESIGN.md section 2. */
static uint32_t wf_class(uint64_t f) {
    uint32_t e = 63u - (uint32_t)__builtin_clzll(f); /* f > 0 */
    uint32_t mant = e >= 2 ? (uint32_t)(f >> (e - 2)) & 3u : (uint32_t)(f << (2 - e)) & 3u;
    return e * 4u + mant;
}

typedef struct wf {
    uint32_t* order; /* nodes with free > 0, sorted by (capacity class desc, index asc) */
    uint64_t* C;     /* C[0] = 0, C[k+1] = sat(C[k] + free[order[k]]) */
    uint32_t cnt;
    uint64_t F;
} wf_t;

static const uint64_t* g_sort_free;
static int wf_cmp(const void* a, const void* b) {
    uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    const uint32_t cx = wf_class(g_sort_free[x]), cy = wf_class(g_sort_free[y]);
    if (cx != cy) return cx > cy ? -1 : 1;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static uint64_t node_free(const uint64_t* cap, const uint8_t* alive, const uint64_t* used, uint32_t j) {
    return (alive[j] && cap[j] > used[j]) ? cap[j] - used[j] : 0;
}

static int wf_build(wf_t* w, const uint64_t* cap, const uint8_t* alive, const uint64_t* used, uint32_t m) {
    uint64_t* fr = (uint64_t*)malloc(((size_t)m + 1) * sizeof(uint64_t));
    w->order = (uint32_t*)malloc(((size_t)m + 1) * sizeof(uint32_t));
    w->C = (uint64_t*)malloc(((size_t)m + 2) * sizeof(uint64_t));
    if (!fr || !w->order || !w->C) return 1;
    w->cnt = 0;
    for (uint32_t j = 0; j < m; ++j) {
        fr[j] = node_free(cap, alive, used, j);
        if (fr[j] > 0) w->order[w->cnt++] = j;
    }
    g_sort_free = fr;
    qsort(w->order, w->cnt, sizeof(uint32_t), wf_cmp);
    w->C[0] = 0;
    for (uint32_t k = 0; k < w->cnt; ++k) {
        uint64_t f = fr[w->order[k]], c = w->C[k];
        w->C[k + 1] = (c + f < c) ? ORC_CAP_INF : c + f; /* saturating */
    }
    w->F = w->C[w->cnt];
    free(fr);
    return 0;
}
static void wf_free(wf_t* w) {
    free(w->order);
    free(w->C);
}
/* Object with exclusive water level Q and load l: the node whose interval [C[k], C[k+1])
 * contains Q takes it iff it fits entirely, Q + l <= C[k+1].  Returns node or NONE. */
static uint32_t wf_place(const wf_t* w, uint64_t Q, uint32_t l) {
    if (w->cnt == 0 || Q >= w->F) return ORC_NONE;
    uint32_t lo = 0, hi = w->cnt; /* largest k with C[k] <= Q */
    while (hi - lo > 1) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (w->C[mid] <= Q) lo = mid; else hi = mid;
    }
    return (Q + (uint64_t)l <= w->C[lo + 1]) ? w->order[lo] : ORC_NONE;
}

/* Spill rounds over `rem` (ordered list of slots); load_of/assign through callbacks is
 * overkill — both callers keep (slot -> object row) in `obj`.  Returns remaining count. */
static uint64_t spill_rounds(uint64_t* rem, uint64_t n_rem, const uint32_t* obj_of, const uint32_t* load,
                             uint32_t* node_of_slot, const uint64_t* cap, const uint8_t* alive,
                             uint64_t* used, uint32_t m, uint32_t rounds, uint32_t* rounds_run) {
    for (uint32_t r = 0; r < rounds && n_rem > 0; ++r) {
        wf_t w;
        if (wf_build(&w, cap, alive, used, m)) return n_rem;
        ++*rounds_run;
        uint64_t Q = 0, keep = 0;
        for (uint64_t t = 0; t < n_rem; ++t) {
            uint64_t slot = rem[t];
            uint32_t row = obj_of ? obj_of[slot] : (uint32_t)slot;
            uint32_t l = load[row];
            uint32_t nd = wf_place(&w, Q, l);
            Q += l;
            if (nd != ORC_NONE) {
                node_of_slot[slot] = nd;
                used[nd] += l; /* takes effect for the NEXT round's free[] only (wf already built) */
            } else {
                rem[keep++] = slot;
            }
        }
        n_rem = keep;
        wf_free(&w);
    }
    return n_rem;
}

/* ---- whole-table solve ------------------------------------------------------------------- */

int orc_tick(const uint32_t* cur, const uint32_t* load, const uint32_t* aff, uint64_t n_obj,
             const uint64_t* cap, const uint8_t* alive, uint32_t m, uint32_t rounds,
             uint32_t* next, uint64_t* used_out, orc_stats* st) {
    return orc_tick_ex(cur, load, aff, n_obj, cap, alive, m, rounds, 0u, next, used_out, st);
}

/* flags & ORC_REF_SELF_ASSIGN (= RIO_GP_CFG_REF_SELF_ASSIGN): a pending row claims its affinity node whether or not
 * membership marks it active, against the node's whole capacity — service.rs:244-252 updates the object onto self.address
 * without asking is_active(self); the eviction of rows on inactive nodes (pass 1) and the water-fill (live nodes only)
 * are unchanged. */
int orc_tick_ex(const uint32_t* cur, const uint32_t* load, const uint32_t* aff, uint64_t n_obj,
                const uint64_t* cap, const uint8_t* alive, uint32_t m, uint32_t rounds, uint32_t flags,
                uint32_t* next, uint64_t* used_out, orc_stats* st) {
    const int self_assign = (flags & ORC_REF_SELF_ASSIGN) != 0;
    orc_stats s;
    memset(&s, 0, sizeof s);
    s.n_objects = n_obj;
    uint64_t* used = used_out;
    uint64_t* run = (uint64_t*)calloc((size_t)m + 1, sizeof(uint64_t));
    uint64_t* fre = (uint64_t*)calloc((size_t)m + 1, sizeof(uint64_t));
    uint8_t* cut = (uint8_t*)calloc((size_t)m + 1, 1);
    uint64_t* rem = (uint64_t*)malloc((size_t)(n_obj + 1) * sizeof(uint64_t));
    uint8_t* kept = (uint8_t*)malloc((size_t)n_obj + 1);
    if (!run || !fre || !cut || !rem || !kept) return 4;
    memset(used, 0, (size_t)m * sizeof(uint64_t));

    /* pass 1 — keep (sticky): placed on an active server stays (service.rs:241-242);
     * placed on an inactive one is evicted (clean_server, service.rs:227-237). */
    for (uint64_t i = 0; i < n_obj; ++i) {
        uint32_t c = cur[i];
        if (c != ORC_NONE && c < m && alive[c]) {
            kept[i] = 1;
            next[i] = c;
            used[c] += load[i];
            ++s.kept;
            s.load_kept += load[i];
        } else {
            kept[i] = 0;
            next[i] = ORC_NONE;
            if (c != ORC_NONE) ++s.evicted;
        }
    }
    /* pass 2 — affinity claim (first touch, service.rs:244-252), capacity-gated: claimants of
     * node a in index order; admitted iff the inclusive prefix of load <= free[a]. */
    for (uint32_t j = 0; j < m; ++j)
        fre[j] = self_assign ? (cap[j] > used[j] ? cap[j] - used[j] : 0) : node_free(cap, alive, used, j);
    uint64_t n_rem = 0;
    for (uint64_t i = 0; i < n_obj; ++i) {
        if (kept[i]) continue;
        uint32_t a = aff[i];
        if (a == ORC_AFF_INACTIVE) { --s.n_objects; continue; } /* not an object (row lifecycle): never placed */
        if (a != ORC_NONE && a < m && (alive[a] || self_assign)) {
            run[a] += load[i];
            if (run[a] <= fre[a]) {
                next[i] = a;
                ++s.claimed;
                s.load_claimed += load[i];
                continue;
            }
            cut[a] = 1;
        }
        rem[n_rem++] = i;
    }
    for (uint32_t j = 0; j < m; ++j) s.cut_nodes += cut[j];
    s.slow_path = (s.cut_nodes > 0 || n_rem > 0) ? 1u : 0u;
    /* admitted claim load per node = sum over admitted claimants; fold into used */
    memset(run, 0, (size_t)m * sizeof(uint64_t));
    for (uint64_t i = 0; i < n_obj; ++i)
        if (!kept[i] && next[i] != ORC_NONE) run[next[i]] += load[i];
    for (uint32_t j = 0; j < m; ++j) used[j] += run[j];

    /* pass 3 — spill: water-fill the rest, index order, onto nodes by (capacity class desc, index asc). */
    uint64_t n_spill0 = n_rem;
    n_rem = spill_rounds(rem, n_rem, NULL, load, next, cap, alive, used, m, rounds, &s.rounds_run);
    for (uint64_t t = 0; t < n_rem; ++t) {
        ++s.unplaced;
        s.load_unplaced += load[rem[t]];
    }
    s.spilled = n_spill0 - n_rem;
    for (uint64_t i = 0; i < n_obj; ++i) s.load_spilled += (!kept[i] && next[i] != ORC_NONE) ? load[i] : 0;
    s.load_spilled -= s.load_claimed;
    if (st) *st = s;
    free(run); free(fre); free(cut); free(rem); free(kept);
    return 0;
}

/* ---- batched get_or_create_placement ----------------------------------------------------- */

int orc_place_pending(uint32_t* assign, const uint32_t* load, uint64_t n_obj, const uint64_t* cap,
                      const uint8_t* alive, uint64_t* used, uint32_t m, uint32_t rounds,
                      const uint32_t* idx, const uint32_t* requester, uint64_t n,
                      uint32_t* out_node, uint32_t* out_flag) {
    return orc_place_pending_ex(assign, load, n_obj, cap, alive, used, m, rounds, 0u, idx, requester, n, out_node, out_flag);
}

/* flags & ORC_REF_SELF_ASSIGN: first touch goes to the requester whether or not it is an active member (see orc_tick_ex) */
int orc_place_pending_ex(uint32_t* assign, const uint32_t* load, uint64_t n_obj, const uint64_t* cap,
                         const uint8_t* alive, uint64_t* used, uint32_t m, uint32_t rounds, uint32_t flags,
                         const uint32_t* idx, const uint32_t* requester, uint64_t n,
                         uint32_t* out_node, uint32_t* out_flag) {
    const int self_assign = (flags & ORC_REF_SELF_ASSIGN) != 0;
    for (uint64_t k = 0; k < n; ++k)
        if (idx[k] >= n_obj || requester[k] >= m) return 1;
    uint64_t words = ((uint64_t)m + 63) / 64;
    uint64_t* dead = (uint64_t*)calloc((size_t)words + 1, sizeof(uint64_t));
    uint8_t* first = (uint8_t*)calloc((size_t)n + 1, 1);   /* first occurrence of its object in the batch */
    uint8_t* state = (uint8_t*)calloc((size_t)n + 1, 1);   /* flag per request */
    uint64_t* rem = (uint64_t*)malloc((size_t)(n + 1) * sizeof(uint64_t));
    uint32_t* slot_node = (uint32_t*)malloc((size_t)(n + 1) * sizeof(uint32_t));
    uint64_t* run = (uint64_t*)calloc((size_t)m + 1, sizeof(uint64_t));
    uint64_t* fre = (uint64_t*)calloc((size_t)m + 1, sizeof(uint64_t));
    uint32_t* seen = (uint32_t*)malloc((size_t)(n_obj + 1) * sizeof(uint32_t));
    if (!dead || !first || !state || !rem || !slot_node || !run || !fre || !seen) return 4;

    /* (1) service.rs:227-237: a requested object placed on an inactive server triggers
     *     clean_server(that server) — ALL of its objects are un-placed. */
    int any_dead = 0;
    uint8_t* wasdead = (uint8_t*)calloc((size_t)n + 1, 1);  /* the request found its object on a server that is not alive */
    if (!wasdead) return 4;
    for (uint64_t k = 0; k < n; ++k) {
        uint32_t c = assign[idx[k]];
        if (c != ORC_NONE && c < m && !alive[c]) { dead[c >> 6] |= 1ull << (c & 63); any_dead = 1; wasdead[k] = 1; }
    }
    if (any_dead) {
        orc_clean_servers(assign, n_obj, dead, m);
        for (uint32_t j = 0; j < m; ++j)
            if ((dead[j >> 6] >> (j & 63)) & 1ull) used[j] = 0;
    }
    /* (2) duplicates: the first request for an object decides, later ones observe. */
    memset(seen, 0xFF, (size_t)n_obj * sizeof(uint32_t));
    for (uint64_t k = 0; k < n; ++k)
        if (seen[idx[k]] == ORC_NONE) { seen[idx[k]] = (uint32_t)k; first[k] = 1; }
    /* (3) first touch on the requester (service.rs:244-252), capacity-gated by the
     *     position-ordered prefix rule; requester must be an active member. */
    for (uint32_t j = 0; j < m; ++j)
        fre[j] = self_assign ? (cap[j] > used[j] ? cap[j] - used[j] : 0) : node_free(cap, alive, used, j);
    uint64_t n_rem = 0;
    for (uint64_t k = 0; k < n; ++k) {
        slot_node[k] = ORC_NONE;
        if (!first[k]) continue;
        uint32_t row = idx[k], c = assign[row], r = requester[k];
        if (c != ORC_NONE) { state[k] = 1; continue; } /* sticky */
        if (alive[r] || self_assign) {
            run[r] += load[row];
            if (run[r] <= fre[r]) { slot_node[k] = r; state[k] = 2; continue; }
        }
        rem[n_rem++] = k;
    }
    for (uint64_t k = 0; k < n; ++k)
        if (state[k] == 2) used[slot_node[k]] += load[idx[k]];
    /* (4) spill in batch order */
    uint32_t rr = 0;
    uint64_t n_rem0 = n_rem;
    (void)n_rem0;
    for (uint64_t t = 0; t < n_rem; ++t) state[rem[t]] = 4;
    n_rem = spill_rounds(rem, n_rem, idx, load, slot_node, cap, alive, used, m, rounds, &rr);
    for (uint64_t k = 0; k < n; ++k)
        if (state[k] == 4 && slot_node[k] != ORC_NONE) state[k] = 3;
    /* commit + outputs */
    for (uint64_t k = 0; k < n; ++k)
        if (first[k] && (state[k] == 2 || state[k] == 3)) assign[idx[k]] = slot_node[k];
    for (uint64_t k = 0; k < n; ++k) {
        uint32_t nd = assign[idx[k]], fl;
        if (first[k] && state[k] == 2) fl = 2;        /* PLACED */
        else if (first[k] && state[k] == 3) fl = 3;   /* SPILLED */
        else if (nd == ORC_NONE) fl = 4;              /* UNPLACED */
        else fl = (nd == requester[k]) ? 0u : 1u;     /* LOCAL / REDIRECT */
        /* service.rs:268-285: placed elsewhere, there is dead -> cleaned and re-placed by this request (the first one
         * for the object; later ones observe the new placement) */
        if (first[k] && wasdead[k]) fl |= ORC_FLAG_REPLACED;
        out_node[k] = nd;
        if (out_flag) out_flag[k] = fl;
    }
    free(dead); free(first); free(state); free(rem); free(slot_node); free(run); free(fre); free(seen); free(wasdead);
    return 0;
}

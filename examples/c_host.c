/* c_host.c — the drop-in boundary driven from plain C (what a cgo / Rust-FFI host does): no Python, no torch.
 *
 * Builds the BASELINE.json config-3 table with the counter-based generator of SURVEY.md §8d (splitmix64; Zipf(1.1) loads by
 * inverse-CDF lookup; the arithmetic is restated from rio-rs_amd/synth.py, so tables agree up to libm rounding),
 * then times, through include/rio_gpu_placement.h only:
 *   fast path   K pipelined whole-table solves of the cold table (rio_gp_solve_async + one rio_gp_solve_wait)
 *   config 5    committed churn ticks: rio_gp_set_alive_all (10 % of the nodes down, others back) + rio_gp_tick
 * and prints one JSON line.  Replaces, for the host side, what Service::get_or_create_placement does per object
 * (rio-rs/src/service.rs:193-254) by one call per tick.
 *
 * Build:  gcc -O2 -std=c99 -I include examples/c_host.c -o examples/c_host -L rio-rs_amd -lrio_gp \
 *             -Wl,-rpath,$PWD/rio-rs_amd -Wl,-rpath,/opt/rocm/lib -lm
 * Run:    examples/c_host [rows=10000000] [nodes=1024] [ticks=100] [solves=200]
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rio_gpu_placement.h"

static const uint64_t SEED = 0x52494F5F52530001ull;

static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static uint64_t r(uint64_t i, uint64_t k) { return splitmix64(SEED ^ (k << 56) ^ i); }
static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
#define CHK(call)                                                                                \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ != RIO_GP_OK) {                                                                  \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rio_gp_last_error(h));                 \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

/* alive[] of churn tick t: the m/10 nodes with the smallest r(t*m + j, 4) are down (ties by index) */
static void churn_mask(uint32_t m, uint64_t tick, uint8_t* alive, uint64_t* score) {
    uint32_t k = m / 10 ? m / 10 : 1, j, d;
    for (j = 0; j < m; ++j) { score[j] = r(tick * m + j, 4); alive[j] = 1; }
    for (d = 0; d < k; ++d) {
        uint32_t best = m;
        for (j = 0; j < m; ++j)
            if (alive[j] && (best == m || score[j] < score[best])) best = j;
        alive[best] = 0;
    }
}

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 10000000ull;
    const uint32_t m = argc > 2 ? (uint32_t)strtoul(argv[2], 0, 10) : 1024u;
    const int ticks = argc > 3 ? atoi(argv[3]) : 100, solves = argc > 4 ? atoi(argv[4]) : 200;
    const int kmax = 65536;
    rio_gp_t* h = 0;
    rio_gp_cfg cfg;
    rio_gp_stats st, last;
    uint32_t *load = malloc(n * 4), *aff = malloc(n * 4), *warm = malloc(n * 4), n_slow = 0;
    uint64_t *cap = malloc((size_t)m * 8), *score = malloc((size_t)m * 8), i, total = 0, moved = 0;
    uint8_t* alive = malloc(m);
    double* cdf = malloc(sizeof(double) * kmax), acc = 0, t0, t_fast, t_churn;
    int k;
    if (!load || !aff || !warm || !cap || !score || !alive || !cdf) return 2;
    for (k = 0; k < kmax; ++k) { acc += pow((double)(k + 1), -1.1); cdf[k] = acc; }
    for (k = 0; k < kmax; ++k) cdf[k] /= acc;
    for (i = 0; i < n; ++i) {
        const double u = (double)(r(i, 3) >> 11) * (1.0 / 9007199254740992.0);
        int lo = 0, hi = kmax; /* searchsorted(cdf, u, side="right") */
        while (lo < hi) { const int mid = (lo + hi) / 2; if (cdf[mid] <= u) lo = mid + 1; else hi = mid; }
        load[i] = (uint32_t)((lo < kmax - 1 ? lo : kmax - 1) + 1);
        aff[i] = (uint32_t)(r(i, 1) % m);
        warm[i] = (uint32_t)(r(i, 2) % m);
        total += load[i];
    }
    for (i = 0; i < m; ++i) { cap[i] = (total * 1250 + 1000ull * m - 1) / (1000ull * m); alive[i] = 1; }

    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (uint32_t)sizeof cfg;
    cfg.max_objects = n;
    cfg.max_nodes = m;
    if (rio_gp_create(&cfg, &h) != RIO_GP_OK) { fprintf(stderr, "rio_gp_create: %s\n", rio_gp_last_error(0)); return 1; }
    CHK(rio_gp_set_nodes(h, m, cap, alive));
    CHK(rio_gp_set_objects(h, n, load, aff));

    /* fast path: cold table, every row pending, pipelined solves, verdicts read once at the end */
    for (k = 0; k < 20; ++k) CHK(rio_gp_solve_async(h));
    CHK(rio_gp_solve_wait(h, &st, &n_slow));
    t0 = now_s();
    for (k = 0; k < solves; ++k) CHK(rio_gp_solve_async(h));
    CHK(rio_gp_solve_wait(h, &st, &n_slow));
    t_fast = now_s() - t0;

    /* config 5: warm table, then one liveness push + one committed tick per step */
    CHK(rio_gp_set_assign(h, n, warm));
    CHK(rio_gp_tick(h, &last));
    for (k = 0; k < 3; ++k) { churn_mask(m, 2 + (uint64_t)k, alive, score); CHK(rio_gp_set_alive_all(h, m, alive)); CHK(rio_gp_tick(h, &last)); }
    t_churn = 0;
    for (k = 0; k < ticks; ++k) {
        churn_mask(m, 5 + (uint64_t)k, alive, score); /* outside the timed region: the host's own bookkeeping */
        t0 = now_s();
        CHK(rio_gp_set_alive_all(h, m, alive));
        CHK(rio_gp_tick(h, &last));
        t_churn += now_s() - t0;
        moved += last.claimed + last.spilled;
    }
    printf("{\"host\": \"C99 over the C ABI\", \"backend\": \"%s\", \"rows\": %llu, \"nodes\": %u, "
           "\"fast_path\": {\"solves\": %d, \"us_per_solve\": %.2f, \"decisions_per_s\": %.4e, \"slow_steps\": %u, \"claimed\": %llu}, "
           "\"churn\": {\"ticks\": %d, \"us_per_tick\": %.2f, \"rows_decided_per_s\": %.4e, \"moved_per_s\": %.4e, "
           "\"last\": {\"kept\": %llu, \"evicted\": %llu, \"claimed\": %llu, \"spilled\": %llu, \"unplaced\": %llu, \"cut_nodes\": %u}}}\n",
           rio_gp_backend(h), (unsigned long long)n, m, solves, t_fast / solves * 1e6, (double)n * solves / t_fast, n_slow,
           (unsigned long long)st.claimed, ticks, t_churn / ticks * 1e6, (double)n * ticks / t_churn, (double)moved / t_churn,
           (unsigned long long)last.kept, (unsigned long long)last.evicted, (unsigned long long)last.claimed,
           (unsigned long long)last.spilled, (unsigned long long)last.unplaced, last.cut_nodes);
    rio_gp_destroy(h);
    free(load); free(aff); free(warm); free(cap); free(score); free(alive); free(cdf);
    return 0;
}

/* c_host_threads.c — the trait-level boundary (include/rio_gpu_object_placement.h) under concurrent callers, as the
 * reference calls it: one task per connection, every request a lookup / get_or_create_placement of ONE object
 * (rio-rs/src/service.rs:193-254, server.rs:292-304).  T threads hammer one shared provider; the library's combining
 * front-end lets concurrent single-object calls share a device round trip.  Prints one JSON line per thread count.
 *
 * Build:  gcc -O2 -std=c99 -pthread -I include examples/c_host_threads.c -o examples/c_host_threads -L rio-rs_amd \
 *             -lrio_gp -Wl,-rpath,$PWD/rio-rs_amd -Wl,-rpath,/opt/rocm/lib
 * Run:    examples/c_host_threads [objects=20000] [calls_per_thread=2000] [max_threads=16]
 * (more calling threads than cores only adds scheduler noise: the box used for profiles/ has a 16-CPU quota)
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rio_gpu_object_placement.h"

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

typedef struct {
    rio_op_t* p;
    int tid, calls, objects, mode; /* mode 0 lookup | 1 get_or_create_placement */
    int bad;
} job;

static void* worker(void* arg) {
    job* j = (job*)arg;
    char id[32], out[64], self[32];
    int k, found;
    uint32_t flag;
    unsigned x = 12345u + 977u * (unsigned)j->tid;
    snprintf(self, sizeof self, "10.0.0.%d:5000", j->tid % 8);
    for (k = 0; k < j->calls; ++k) {
        x = x * 1664525u + 1013904223u;
        snprintf(id, sizeof id, "%u", (x >> 8) % (unsigned)j->objects);
        if (j->mode == 0) {
            char want[32];
            if (rio_op_lookup(j->p, "Obj", id, out, sizeof out, &found) != RIO_GP_OK) { j->bad++; continue; }
            snprintf(want, sizeof want, "10.0.0.%u:5000", (unsigned)atoi(id) % 8u);
            if (!found || strcmp(out, want) != 0) j->bad++;
        } else {
            if (rio_op_get_or_create_placement(j->p, "Obj", id, self, out, sizeof out, &flag) != RIO_GP_OK) j->bad++;
            else if (!out[0]) j->bad++; /* capacity is unbounded: every object ends up somewhere */
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    const int objects = argc > 1 ? atoi(argv[1]) : 20000, calls = argc > 2 ? atoi(argv[2]) : 2000;
    const int max_threads = argc > 3 ? atoi(argv[3]) : 16;
    const int counts[] = {1, 4, 16, 64, 256};
    rio_op_cfg cfg;
    rio_op_t* p = 0;
    int i, c, mode;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (uint32_t)sizeof cfg;
    cfg.max_objects = (uint64_t)objects * 2;
    cfg.max_nodes = 64;
    if (rio_op_create(&cfg, &p) != RIO_GP_OK) { fprintf(stderr, "rio_op_create: %s\n", rio_op_last_error(0)); return 1; }
    for (i = 0; i < 8; ++i) {
        char a[32];
        snprintf(a, sizeof a, "10.0.0.%d:5000", i);
        if (rio_op_set_member(p, a, 1, RIO_GP_CAP_INF) != RIO_GP_OK) return 1;
    }
    { /* objects 0..objects-1 placed on node (i mod 8) in one batched update */
        const char** ty = malloc(sizeof(char*) * (size_t)objects);
        const char** id = malloc(sizeof(char*) * (size_t)objects);
        const char** ad = malloc(sizeof(char*) * (size_t)objects);
        char* ids = malloc((size_t)objects * 16), *ads = malloc((size_t)objects * 24);
        for (i = 0; i < objects; ++i) {
            snprintf(ids + (size_t)i * 16, 16, "%d", i);
            snprintf(ads + (size_t)i * 24, 24, "10.0.0.%d:5000", i % 8);
            ty[i] = "Obj"; id[i] = ids + (size_t)i * 16; ad[i] = ads + (size_t)i * 24;
        }
        if (rio_op_update_batch(p, (uint64_t)objects, ty, id, ad) != RIO_GP_OK) { fprintf(stderr, "%s\n", rio_op_last_error(p)); return 1; }
        free(ty); free(id); free(ad); free(ids); free(ads);
    }
    for (mode = 0; mode < 2; ++mode)
        for (c = 0; c < (int)(sizeof counts / sizeof counts[0]); ++c) {
            const int T = counts[c];
            if (T > max_threads) continue;
            pthread_t* th = malloc(sizeof(pthread_t) * (size_t)T);
            job* jobs = malloc(sizeof(job) * (size_t)T);
            double t0, dt;
            int bad = 0;
            for (i = 0; i < T; ++i) { jobs[i].p = p; jobs[i].tid = i; jobs[i].calls = calls; jobs[i].objects = objects; jobs[i].mode = mode; jobs[i].bad = 0; }
            t0 = now_s();
            for (i = 0; i < T; ++i) pthread_create(&th[i], 0, worker, &jobs[i]);
            for (i = 0; i < T; ++i) { pthread_join(th[i], 0); bad += jobs[i].bad; }
            dt = now_s() - t0;
            printf("{\"call\": \"%s\", \"threads\": %d, \"calls\": %d, \"calls_per_s\": %.4e, \"us_per_call_per_thread\": %.2f, \"wrong\": %d}\n",
                   mode ? "get_or_create_placement" : "lookup", T, T * calls, (double)T * calls / dt, dt / calls * 1e6, bad);
            free(th); free(jobs);
            if (bad) { rio_op_release(p); return 3; }
        }
    rio_op_release(p);
    return 0;
}

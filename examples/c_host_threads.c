/* c_host_threads.c — the trait-level boundary (include/rio_gpu_object_placement.h) under concurrent callers, as the
 * reference calls it: one task per connection, every request a lookup / get_or_create_placement of ONE object
 * (rio-rs/src/service.rs:193-254, server.rs:292-304).  T threads hammer one shared provider.  Prints one JSON line per
 * (provider, call, thread count):
 *   provider "shadow"     the default: answers the device has given are remembered on the host, a hit costs no round trip
 *            "device"     RIO_OP_CFG_NO_HOST_SHADOW: every call goes to the device; concurrent callers share round trips
 *                         (flat combining)
 *   call     lookup                    known, placed keys (local.rs:42-49)
 *            get_or_create_placement   sticky hits from any of 8 servers (service.rs:199-242)
 *            churn                     9 lookups, then remove + get_or_create_placement of one object of the caller's own
 *                                      (a first touch: always the device) — the mix a server with ~10 % activations sees
 *   entry    direct     the caller's thread makes the blocking call itself (pthreads on rio_op_lookup: what round 5 measured)
 *            pool       EVERY call is handed to a pool of blocking threads and the caller waits for its completion — what an async
 *                       host pays when each trait method sits behind tokio::task::spawn_blocking (a queue, a wake-up, a wake-up back)
 *            try        rio_op_try_* inline on the caller's thread — the host shadow or RIO_GP_EAGAIN, never the device —, and only on
 *                       EAGAIN the hand-off to the pool: what rio-rs_amd/rust/src/gpu.rs does on the async worker
 *
 * Build:  gcc -O2 -std=c99 -pthread -I include examples/c_host_threads.c -o examples/c_host_threads -L rio-rs_amd \
 *             -lrio_gp -Wl,-rpath,$PWD/rio-rs_amd -Wl,-rpath,/opt/rocm/lib
 * Run:    examples/c_host_threads [objects=20000] [calls_per_thread=2000] [max_threads=256] [collect_ns=0 (default)]
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
    int tid, calls, objects, mode; /* mode 0 lookup | 1 get_or_create_placement | 2 churn */
    int entry;                     /* 0 direct | 1 pool | 2 try, then pool */
    int bad;
    long tried, answered;          /* entry 2: rio_op_try_* calls made / answered from the shadow */
} job;

/* ---- a pool of blocking threads (the stand-in for tokio's spawn_blocking): tasks = one blocking trait call each ---- */
typedef struct task {
    rio_op_t* p;
    int kind;                      /* 0 lookup | 1 get_or_create_placement | 3 remove */
    const char *ty, *id, *self;
    char* out; size_t cap; int* found; uint32_t* flag;
    int rc, done;
    pthread_mutex_t mu; pthread_cond_t cv;
    struct task* next;
} task;
static struct { pthread_mutex_t mu; pthread_cond_t cv; task *head, *tail; int stop, n; pthread_t th[64]; } g_pool;
static void* pool_main(void* arg) {
    (void)arg;
    for (;;) {
        task* t;
        pthread_mutex_lock(&g_pool.mu);
        while (!g_pool.head && !g_pool.stop) pthread_cond_wait(&g_pool.cv, &g_pool.mu);
        if (!g_pool.head) { pthread_mutex_unlock(&g_pool.mu); return 0; }
        t = g_pool.head; g_pool.head = t->next; if (!g_pool.head) g_pool.tail = 0;
        pthread_mutex_unlock(&g_pool.mu);
        t->rc = t->kind == 0 ? rio_op_lookup(t->p, t->ty, t->id, t->out, t->cap, t->found)
              : t->kind == 1 ? rio_op_get_or_create_placement(t->p, t->ty, t->id, t->self, t->out, t->cap, t->flag)
                             : rio_op_remove(t->p, t->ty, t->id);
        pthread_mutex_lock(&t->mu); t->done = 1; pthread_cond_signal(&t->cv); pthread_mutex_unlock(&t->mu);
    }
}
static void pool_start(int n) {
    int i;
    memset(&g_pool, 0, sizeof g_pool);
    pthread_mutex_init(&g_pool.mu, 0); pthread_cond_init(&g_pool.cv, 0);
    g_pool.n = n > 64 ? 64 : n;
    for (i = 0; i < g_pool.n; ++i) pthread_create(&g_pool.th[i], 0, pool_main, 0);
}
static void pool_stop(void) {
    int i;
    pthread_mutex_lock(&g_pool.mu); g_pool.stop = 1; pthread_cond_broadcast(&g_pool.cv); pthread_mutex_unlock(&g_pool.mu);
    for (i = 0; i < g_pool.n; ++i) pthread_join(g_pool.th[i], 0);
}
static int pool_call(task* t) { /* enqueue, wake a pool thread, wait for the completion */
    t->done = 0; t->next = 0;
    pthread_mutex_init(&t->mu, 0); pthread_cond_init(&t->cv, 0);
    pthread_mutex_lock(&g_pool.mu);
    if (g_pool.tail) g_pool.tail->next = t; else g_pool.head = t;
    g_pool.tail = t;
    pthread_cond_signal(&g_pool.cv);
    pthread_mutex_unlock(&g_pool.mu);
    pthread_mutex_lock(&t->mu);
    while (!t->done) pthread_cond_wait(&t->cv, &t->mu);
    pthread_mutex_unlock(&t->mu);
    pthread_mutex_destroy(&t->mu); pthread_cond_destroy(&t->cv);
    return t->rc;
}
/* the three calls of a worker through the chosen entry */
static int do_lookup(job* j, const char* ty, const char* id, char* out, size_t cap, int* found) {
    if (j->entry == 2) {
        int rc;
        j->tried++;
        rc = rio_op_try_lookup_n(j->p, ty, strlen(ty), id, strlen(id), out, cap, found);
        if (rc != RIO_GP_EAGAIN) { j->answered++; return rc; }
    }
    if (j->entry == 0) return rio_op_lookup(j->p, ty, id, out, cap, found);
    { task t; memset(&t, 0, sizeof t); t.p = j->p; t.kind = 0; t.ty = ty; t.id = id; t.out = out; t.cap = cap; t.found = found; return pool_call(&t); }
}
static int do_request(job* j, const char* ty, const char* id, const char* self, char* out, size_t cap, uint32_t* flag) {
    if (j->entry == 2) {
        int rc;
        j->tried++;
        rc = rio_op_try_get_or_create_placement_n(j->p, ty, strlen(ty), id, strlen(id), self, out, cap, flag);
        if (rc != RIO_GP_EAGAIN) { j->answered++; return rc; }
    }
    if (j->entry == 0) return rio_op_get_or_create_placement(j->p, ty, id, self, out, cap, flag);
    { task t; memset(&t, 0, sizeof t); t.p = j->p; t.kind = 1; t.ty = ty; t.id = id; t.self = self; t.out = out; t.cap = cap; t.flag = flag; return pool_call(&t); }
}
static int do_remove(job* j, const char* ty, const char* id) {
    if (j->entry == 0) return rio_op_remove(j->p, ty, id);
    { task t; memset(&t, 0, sizeof t); t.p = j->p; t.kind = 3; t.ty = ty; t.id = id; return pool_call(&t); }
}

static void* worker(void* arg) {
    job* j = (job*)arg;
    char id[32], out[64], self[32], own[32];
    int k, found;
    uint32_t flag;
    unsigned x = 12345u + 977u * (unsigned)j->tid;
    snprintf(self, sizeof self, "10.0.0.%d:5000", j->tid % 8);
    snprintf(own, sizeof own, "own%d", j->tid);
    for (k = 0; k < j->calls; ++k) {
        x = x * 1664525u + 1013904223u;
        snprintf(id, sizeof id, "%u", (x >> 8) % (unsigned)j->objects);
        if (j->mode == 2 && k % 10 == 9) { /* an activation: the object is not placed, the request first-touches it */
            if (do_remove(j, "Own", own) != RIO_GP_OK) j->bad++;
            if (do_request(j, "Own", own, self, out, sizeof out, &flag) != RIO_GP_OK) j->bad++;
            else if (strcmp(out, self) != 0 || (flag & RIO_GP_FLAG_MASK) != RIO_GP_FLAG_PLACED) j->bad++;
        } else if (j->mode != 1) {
            char want[32];
            if (do_lookup(j, "Obj", id, out, sizeof out, &found) != RIO_GP_OK) { j->bad++; continue; }
            snprintf(want, sizeof want, "10.0.0.%u:5000", (unsigned)atoi(id) % 8u);
            if (!found || strcmp(out, want) != 0) j->bad++;
        } else {
            char want[32];
            if (do_request(j, "Obj", id, self, out, sizeof out, &flag) != RIO_GP_OK) { j->bad++; continue; }
            snprintf(want, sizeof want, "10.0.0.%u:5000", (unsigned)atoi(id) % 8u);
            if (strcmp(out, want) != 0 || flag != (strcmp(want, self) == 0 ? RIO_GP_FLAG_LOCAL : RIO_GP_FLAG_REDIRECT)) j->bad++;
        }
    }
    return 0;
}

static uint32_t g_collect_ns = 0;
static rio_op_t* provider(int objects, uint32_t flags) {
    rio_op_cfg cfg;
    rio_op_t* p = 0;
    int i;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (uint32_t)sizeof cfg;
    cfg.max_objects = (uint64_t)objects * 2 + 1024;
    cfg.max_nodes = 64;
    cfg.flags = flags;
    cfg.collect_ns = g_collect_ns;
    if (rio_op_create(&cfg, &p) != RIO_GP_OK) { fprintf(stderr, "rio_op_create: %s\n", rio_op_last_error(0)); return 0; }
    for (i = 0; i < 8; ++i) {
        char a[32];
        snprintf(a, sizeof a, "10.0.0.%d:5000", i);
        if (rio_op_set_member(p, a, 1, RIO_GP_CAP_INF) != RIO_GP_OK) return 0;
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
        if (rio_op_update_batch(p, (uint64_t)objects, ty, id, ad) != RIO_GP_OK) { fprintf(stderr, "%s\n", rio_op_last_error(p)); return 0; }
        free(ty); free(id); free(ad); free(ids); free(ads);
    }
    return p;
}

int main(int argc, char** argv) {
    const int objects = argc > 1 ? atoi(argv[1]) : 20000, calls = argc > 2 ? atoi(argv[2]) : 2000;
    const int max_threads = argc > 3 ? atoi(argv[3]) : 256;
    const int counts[] = {1, 4, 16, 64, 256};
    const char* names[] = {"lookup", "get_or_create_placement", "churn"};
    const char* entries[] = {"direct", "pool", "try"};
    int i, c, mode, prov, entry;
    g_collect_ns = argc > 4 ? (uint32_t)atoi(argv[4]) : 0u; /* 0 = the library's default, 1 = no collect window */
    pool_start(16);                                          /* (as many blocking threads as the box grants CPUs) */
    for (prov = 0; prov < 2; ++prov) {
        rio_op_t* p = provider(objects, prov ? RIO_OP_CFG_NO_HOST_SHADOW : 0u);
        if (!p) return 1;
        for (entry = 0; entry < (prov ? 1 : 3); ++entry)     /* (without a shadow every rio_op_try_* call is EAGAIN: direct only) */
        for (mode = 0; mode < 3; ++mode)
            for (c = 0; c < (int)(sizeof counts / sizeof counts[0]); ++c) {
                const int T = counts[c];
                uint64_t b0 = 0, r0 = 0, b1 = 0, r1 = 0;
                pthread_t* th;
                job* jobs;
                double t0, dt;
                int bad = 0;
                long tried = 0, answered = 0;
                if (T > max_threads) continue;
                if (entry && (T == 4 || T == 256)) continue;  /* (the hand-off rows: 1, 16 and 64 callers) */
                th = malloc(sizeof(pthread_t) * (size_t)T);
                jobs = malloc(sizeof(job) * (size_t)T);
                /* calls the shadow answers take a fraction of a microsecond: ten times as many of them, or thread start-up is what gets timed */
                const int ncalls = (prov == 0 && mode < 2 && entry != 1) ? calls * 10 : calls;
                for (i = 0; i < T; ++i) { jobs[i].p = p; jobs[i].tid = i; jobs[i].calls = ncalls; jobs[i].objects = objects; jobs[i].mode = mode;
                                          jobs[i].entry = entry; jobs[i].bad = 0; jobs[i].tried = 0; jobs[i].answered = 0; }
                { /* a container with a CPU quota (cgroup cpu.max) throttles the whole process once a scheduler period's allowance
                     is spent: start every configuration in a fresh period, not in the debt of the one before */
                    const struct timespec nap = {0, 150000000};
                    nanosleep(&nap, 0);
                }
                rio_op_device_round_trips(p, &b0, &r0);
                t0 = now_s();
                for (i = 0; i < T; ++i) pthread_create(&th[i], 0, worker, &jobs[i]);
                for (i = 0; i < T; ++i) { pthread_join(th[i], 0); bad += jobs[i].bad; tried += jobs[i].tried; answered += jobs[i].answered; }
                dt = now_s() - t0;
                rio_op_device_round_trips(p, &b1, &r1);
                printf("{\"provider\": \"%s\", \"entry\": \"%s\", \"call\": \"%s\", \"threads\": %d, \"calls\": %d, \"calls_per_s\": %.4e, "
                       "\"us_per_call_per_thread\": %.2f, \"device_round_trips\": %llu, \"requests_on_device\": %llu, \"try_calls\": %ld, "
                       "\"try_answered\": %ld, \"collect_ns\": %u, \"wrong\": %d}\n",
                       prov ? "device" : "shadow", entries[entry], names[mode], T, T * ncalls, (double)T * ncalls / dt, dt / ncalls * 1e6,
                       (unsigned long long)(b1 - b0), (unsigned long long)(r1 - r0), tried, answered,
                       g_collect_ns ? g_collect_ns : RIO_OP_DEFAULT_COLLECT_NS, bad);
                fflush(stdout);
                free(th); free(jobs);
                if (bad) { rio_op_release(p); return 3; }
            }
        rio_op_release(p);
    }
    pool_stop();
    return 0;
}

/*
 * cpu_baseline.c -- multi-threaded CPU timing harness for bench.py's `cpu_baseline` leg.
 *
 * TEST/BENCH INFRASTRUCTURE ONLY (never linked into the product).
 *
 * kind "reference": loads oracle/_ref/libref_cordic_<PW>_<W>.so -- cordic() of the reference's
 *   cpp/cordic_sincos.cpp compiled from its own source (oracle/Makefile) -- and evaluates the window
 *   the way the reference's models do: K-1 calls of cordic(k*n mod N) per coefficient
 *   (hls/windows/win_function.cpp:361-366) followed by the HLS cosine-sum (:368-375).
 *   The reference function recomputes its rescaled ROM on every call (cpp/cordic_sincos.cpp:15-18);
 *   that cost is part of the reference and is kept.
 * kind "port": the same sweep through the oracle's restatement (bhwo_generate).
 * Threads take contiguous index shards; the return value is wall seconds.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bhw_oracle.h"

typedef void (*ref_cordic_fn)(int theta, long long *lut, int *s, int *c);

typedef struct {
    ref_cordic_fn fn;
    const bhwo_params *p;
    uint64_t n0, count;
    int32_t *out;
    long long lut[48];
} job_t;

static void *ref_worker(void *arg)
{
    job_t *j = (job_t *)arg;
    const unsigned W = j->p->dat_width, K = j->p->n_terms;
    const uint64_t mask = (1ull << j->p->phi_width) - 1ull;
    for (uint64_t i = 0; i < j->count; ++i) {
        const uint64_t n = (j->n0 + i) & mask;
        long long acc = j->p->aa[0];
        for (unsigned k = 1; k < K; ++k) {
            int s, c;
            j->fn((int)((k * n) & mask), j->lut, &s, &c);
            long long m = ((long long)j->p->aa[k] * (long long)c) >> (W - 2);
            acc += (k & 1) ? -m : m;
        }
        if (W < 32) { unsigned sh = 64 - W; acc = (long long)((unsigned long long)acc << sh) >> sh; }
        j->out[i] = (int32_t)acc;
    }
    return NULL;
}

static void *port_worker(void *arg)
{
    job_t *j = (job_t *)arg;
    bhwo_generate(j->p, j->n0, j->count, j->out);
    return NULL;
}

typedef struct {
    const bhwo_params *p;
    uint64_t theta0, count;
    int32_t *s, *c;
} sc_job_t;

static void *sincos_worker(void *arg)
{
    sc_job_t *j = (sc_job_t *)arg;
    bhwo_sincos(j->p, j->theta0, j->count, j->s, j->c);
    return NULL;
}

/* bhwo_sincos over host threads (contiguous phase shards): whole-quadrant sweeps in tests.  Returns 0 / -1. */
int bhw_cpu_sincos_mt(const bhwo_params *p, uint64_t theta0, uint64_t count, int threads, int32_t *out_sin, int32_t *out_cos)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    int32_t probe_s, probe_c;
    if (bhwo_sincos(p, theta0, 1, &probe_s, &probe_c)) return -1;
    sc_job_t *jobs = (sc_job_t *)calloc((size_t)threads, sizeof(sc_job_t));
    pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    const uint64_t base = count / (uint64_t)threads, rem = count % (uint64_t)threads;
    uint64_t off = 0;
    for (int t = 0; t < threads; ++t) {
        jobs[t].p = p;
        jobs[t].theta0 = theta0 + off;
        jobs[t].count = base + ((uint64_t)t < rem ? 1 : 0);
        jobs[t].s = out_sin + off;
        jobs[t].c = out_cos + off;
        off += jobs[t].count;
        pthread_create(&tid[t], NULL, sincos_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
    free(jobs);
    free(tid);
    return 0;
}

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ref_lib == NULL -> "port".  Returns seconds, or a negative value on error. */
double bhw_cpu_baseline(const char *ref_lib, const bhwo_params *p, uint64_t n0, uint64_t count, int threads, int32_t *out)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    ref_cordic_fn fn = NULL;
    void *h = NULL;
    if (ref_lib) {
        h = dlopen(ref_lib, RTLD_NOW | RTLD_LOCAL);
        if (!h) return -1.0;
        fn = (ref_cordic_fn)dlsym(h, "_Z6cordiciPxPiS0_");
        if (!fn) { dlclose(h); return -2.0; }
    }
    job_t *jobs = (job_t *)calloc((size_t)threads, sizeof(job_t));
    pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    const int64_t *t2 = bhwo_table_t2();
    const uint64_t base = count / (uint64_t)threads, rem = count % (uint64_t)threads;
    uint64_t off = 0;
    const double t0 = now();
    for (int t = 0; t < threads; ++t) {
        jobs[t].fn = fn;
        jobs[t].p = p;
        jobs[t].n0 = n0 + off;
        jobs[t].count = base + ((uint64_t)t < rem ? 1 : 0);
        jobs[t].out = out + off;
        for (int i = 0; i < 48; ++i) jobs[t].lut[i] = t2[i];
        off += jobs[t].count;
        pthread_create(&tid[t], NULL, fn ? ref_worker : port_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
    const double dt = now() - t0;
    free(jobs);
    free(tid);
    if (h) dlclose(h);
    return dt;
}

/* heapguard: an LD_PRELOAD quarantine for free() that finds WRITES INTO FREED HOST MEMORY, whoever makes them.
 *
 *     gcc -O2 -g -shared -fPIC -o libheapguard.so heapguard.c -ldl -lpthread
 *     HEAPGUARD_LOG=gpurun_out/heapguard.log PYTHONMALLOC=malloc LD_PRELOAD=$PWD/tools/heapguard/libheapguard.so python3 -m pytest ...
 *
 * Why: twice in this build a byte of a freshly allocated copy of a program text changed between
 * its memcpy and its parse.  Memory that has just been allocated is memory somebody has just freed, so the likeliest writer holds
 * a pointer into a block it (or someone else) freed -- a C++ reference into a reallocated vector, a late copy, a runtime thread.
 * The symptom needs the stray write to land on the few hundred bytes being parsed at that moment; the write itself may happen in
 * every run.  This tool makes every such write visible: free() does not hand the block back, it fills it with 0xD5 and parks it in
 * a ring; when the block leaves the ring (and at exit, and on demand through heapguard_sweep()) every byte must still be 0xD5.
 * A block that is not is reported with its size, who freed it (module + offset of up to 10 frames: `addr2line -e <module> <offset>`),
 * how long ago, and which bytes now hold what.  (One frame by default -- the caller of free / operator delete; HEAPGUARD_TRACE=1
 * asks the unwinder for ten, which can deadlock in programs that throw.)
 *
 * Only free() is interposed (malloc / realloc / memalign stay glibc's, so there is no bootstrap problem and any pointer glibc hands out
 * can be parked).  Blocks above HEAPGUARD_MAX_BLOCK (default 256 KiB; mmap-ed chunks) or below HEAPGUARD_MIN_BLOCK (default 0) go straight
 * back; HEAPGUARD_RING entries (default 65536) and HEAPGUARD_BYTES (default 512 MiB) bound the ring: the longer a block stays, the likelier
 * a late write finds it still parked.  Test tooling only: nothing in
 * the product links or loads it. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <malloc.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define FILL 0xD5
#define NFRAMES 10

typedef struct {
    void *p;
    size_t size;
    void *site[NFRAMES];
    int nsite;
    double t_free;
    long tid;
} parked;

static void (*real_free)(void *);
static parked *ring;
static size_t ring_cap = 1u << 16, ring_head, ring_count;
static size_t bytes_parked, bytes_cap = (size_t)512 << 20, max_block = (size_t)256 << 10, min_block = 0;
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static __thread int inside;
static int ready, disabled, deep_trace;
static long n_parked, n_violations;
static FILE *logf;

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void describe_frame(FILE *f, void *a) {
    Dl_info di;
    if (dladdr(a, &di) && di.dli_fname) fprintf(f, "      %s +0x%lx%s%s\n", di.dli_fname, (unsigned long)((char *)a - (char *)di.dli_fbase), di.dli_sname ? "  " : "", di.dli_sname ? di.dli_sname : "");
    else fprintf(f, "      %p\n", a);
}

static void report(const parked *e, const char *when) {
    const unsigned char *b = (const unsigned char *)e->p;
    size_t first = 0, last = 0, nbad = 0;
    for (size_t i = 0; i < e->size; i++)
        if (b[i] != FILL) { if (!nbad) first = i; last = i; nbad++; }
    n_violations++;
    FILE *outs[2] = {stderr, logf};
    for (int k = 0; k < 2; k++) {
        FILE *f = outs[k];
        if (!f) continue;
        fprintf(f, "[heapguard] WRITE AFTER FREE (%s): block %p of %zu bytes, freed %.6f s ago by thread %ld; %zu byte(s) changed, offsets %zu..%zu\n",
                when, e->p, e->size, now() - e->t_free, e->tid, nbad, first, last);
        fprintf(f, "    bytes from offset %zu:", first);
        for (size_t i = first; i < e->size && i < first + 48; i++) fprintf(f, " %02x", b[i]);
        fprintf(f, "\n    as text: \"");
        for (size_t i = first; i < e->size && i < first + 48; i++) fputc(b[i] >= 32 && b[i] < 127 ? b[i] : '.', f);
        fprintf(f, "\"\n    freed from:\n");
        for (int i = 0; i < e->nsite; i++) describe_frame(f, e->site[i]);
        fflush(f);
    }
}

static void release(parked *e, const char *when) {
    const unsigned char *b = (const unsigned char *)e->p;
    size_t i = 0;
    /* word-wise check first */
    const uint64_t want = 0xD5D5D5D5D5D5D5D5ull;
    int bad = 0;
    for (; i + 8 <= e->size; i += 8) { uint64_t w; memcpy(&w, b + i, 8); if (w != want) { bad = 1; break; } }
    if (!bad) for (; i < e->size; i++) if (b[i] != FILL) { bad = 1; break; }
    if (bad) report(e, when);
    bytes_parked -= e->size;
    real_free(e->p);
    e->p = NULL;
}

static void at_exit_sweep(void);

static void init(void) {
    inside = 1;
    real_free = (void (*)(void *))dlsym(RTLD_NEXT, "free");
    const char *s;
    if ((s = getenv("HEAPGUARD_RING"))) ring_cap = (size_t)strtoull(s, NULL, 10);
    if ((s = getenv("HEAPGUARD_MAX_BLOCK"))) max_block = (size_t)strtoull(s, NULL, 10);
    if ((s = getenv("HEAPGUARD_MIN_BLOCK"))) min_block = (size_t)strtoull(s, NULL, 10);
    if ((s = getenv("HEAPGUARD_BYTES"))) bytes_cap = (size_t)strtoull(s, NULL, 10);
    if ((s = getenv("HEAPGUARD_OFF")) && *s == '1') disabled = 1;
    if ((s = getenv("HEAPGUARD_TRACE")) && *s == '1') deep_trace = 1;
    ring = (parked *)calloc(ring_cap, sizeof(parked));
    if ((s = getenv("HEAPGUARD_LOG")) && *s) {
        char path[4096];
        snprintf(path, sizeof path, "%s.%d", s, (int)getpid());
        logf = fopen(path, "w");
    }
    void *warm[4];
    backtrace(warm, 4);                              /* loads libgcc_s now, not inside somebody's free() */
    atexit(at_exit_sweep);
    ready = 1;
    inside = 0;
}

/* every parked block checked where it is (nothing released): returns the number of violations seen so far */
long heapguard_sweep(void) {
    if (!ready || disabled) return n_violations;
    inside++;
    pthread_mutex_lock(&mu);
    for (size_t k = 0; k < ring_count; k++) {
        parked *e = &ring[(ring_head + k) % ring_cap];
        if (!e->p) continue;
        const unsigned char *b = (const unsigned char *)e->p;
        for (size_t i = 0; i < e->size; i++)
            if (b[i] != FILL) { report(e, "sweep"); memset(e->p, FILL, e->size); break; }
    }
    pthread_mutex_unlock(&mu);
    inside--;
    return n_violations;
}

long heapguard_parked(void) { return n_parked; }

static void at_exit_sweep(void) {
    heapguard_sweep();
    FILE *outs[2] = {stderr, logf};
    for (int k = 0; k < 2; k++)
        if (outs[k] && (k == 1 || !logf || n_violations)) { fprintf(outs[k], "[heapguard] pid %d: %ld blocks parked in all, %ld write-after-free report(s)\n", (int)getpid(), n_parked, n_violations); fflush(outs[k]); }
    disabled = 1;                                    /* the process is going down: frees pass through from here */
}

static void park(void *p, void *site) {
    if (!p) return;
    if (!ready) {
        if (inside) return;                          /* dlsym's own free during init: a leak of a few bytes */
        static pthread_once_t once = PTHREAD_ONCE_INIT;
        pthread_once(&once, init);
    }
    if (inside || disabled) { real_free(p); return; }
    size_t size = malloc_usable_size(p);
    if (size == 0 || size > max_block || size < min_block) { real_free(p); return; }
    inside++;
    parked e;
    e.p = p; e.size = size; e.t_free = now(); e.tid = (long)pthread_self();
    /* the immediate caller always; the unwinder only on request (HEAPGUARD_TRACE=1): it takes libgcc's object lock, which a free()
     * made while an exception is being thrown or a DSO's frames are being deregistered already holds -- a self-deadlock */
    e.site[0] = site; e.nsite = 1;
    if (deep_trace && size >= 128 && size <= 16384) e.nsite = backtrace(e.site, NFRAMES);
    memset(p, FILL, size);
    pthread_mutex_lock(&mu);
    while (ring_count == ring_cap || (ring_count && bytes_parked + size > bytes_cap)) {
        parked *old = &ring[ring_head];
        ring_head = (ring_head + 1) % ring_cap;
        ring_count--;
        if (old->p) release(old, "leaving the ring");
    }
    ring[(ring_head + ring_count) % ring_cap] = e;
    ring_count++;
    bytes_parked += size;
    n_parked++;
    pthread_mutex_unlock(&mu);
    inside--;
}

void free(void *p) { park(p, __builtin_return_address(0)); }
/* C++ deletes (libstdc++'s go to free(): interposed here so that the recorded site is the caller of `delete`, not libstdc++) */
void _ZdlPv(void *p) { park(p, __builtin_return_address(0)); }
void _ZdaPv(void *p) { park(p, __builtin_return_address(0)); }
void _ZdlPvm(void *p, unsigned long n) { (void)n; park(p, __builtin_return_address(0)); }
void _ZdaPvm(void *p, unsigned long n) { (void)n; park(p, __builtin_return_address(0)); }
void _ZdlPvSt11align_val_t(void *p, unsigned long a) { (void)a; park(p, __builtin_return_address(0)); }
void _ZdlPvmSt11align_val_t(void *p, unsigned long n, unsigned long a) { (void)n; (void)a; park(p, __builtin_return_address(0)); }
void _ZdaPvSt11align_val_t(void *p, unsigned long a) { (void)a; park(p, __builtin_return_address(0)); }
void _ZdaPvmSt11align_val_t(void *p, unsigned long n, unsigned long a) { (void)n; (void)a; park(p, __builtin_return_address(0)); }

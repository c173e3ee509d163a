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
 * Round 4: two more classes of stray writes.
 *   RED ZONES (HEAPGUARD_REDZONE=1): malloc / calloc / realloc / memalign / posix_memalign / aligned_alloc are interposed as well; every
 *   block is 32 bytes longer than asked for: 8 canary bytes right behind the user's last byte, and at the very end of the block a trailer
 *   {size ^ magic, allocating site, magic}.  free() / realloc() check the canary: a store ONE BYTE PAST THE END OF A LIVE BLOCK -- which
 *   the quarantine cannot see, and which lands in the neighbouring block when there is no red zone -- is reported with the block's size
 *   and who allocated it.  (Blocks handed out before this library was ready carry no trailer and are passed through.)
 *   PINNED HOST MEMORY: hipHostMalloc / hipHostFree / hipHostRegister / hipHostUnregister are interposed: every host-visible range is
 *   logged with its lifetime (HEAPGUARD_LOG), and hipHostFree quarantines like free(): the range is filled with 0xD5 and only given back
 *   to the runtime after HEAPGUARD_PINNED_RING later frees (default 64) or at exit, when it must still hold the fill -- a kernel or a DMA
 *   copy that still writes a released pinned range never passes through free() and was invisible before.
 *
 * free() is always interposed (any pointer glibc hands out can be parked); the allocating calls only with HEAPGUARD_REDZONE=1.  Blocks above HEAPGUARD_MAX_BLOCK (default 256 KiB; mmap-ed chunks) or below HEAPGUARD_MIN_BLOCK (default 0) go straight
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
static __thread int inside __attribute__((tls_model("initial-exec")));     /* (initial-exec: the lazy TLS of a shared object is allocated with malloc -- which is interposed here) */
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

static void pin_sweep_at_exit(void);
static long n_overruns, n_zoned, n_pin_allocs, n_pin_violations;
static void at_exit_sweep(void) {
    heapguard_sweep();
    pin_sweep_at_exit();
    FILE *outs[2] = {stderr, logf};
    for (int k = 0; k < 2; k++)
        if (outs[k] && (k == 1 || !logf || n_violations || n_overruns || n_pin_violations)) { fprintf(outs[k], "[heapguard] pid %d: %ld blocks parked in all, %ld write-after-free report(s); %ld blocks with a red zone, %ld overrun report(s); %ld pinned ranges, %ld write(s) into released pinned memory\n", (int)getpid(), n_parked, n_violations, n_zoned, n_overruns, n_pin_allocs, n_pin_violations); fflush(outs[k]); }
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

/* ---- red zones --------------------------------------------------------------------------------------------------------------- */
#define RZ_CANARY 0xC5C5C5C5C5C5C5C5ull
#define RZ_MAGIC1 0x5A17ED0C0FFEE123ull
#define RZ_MAGIC2 0x0DDBA11CAFEF00D5ull
#define RZ_EXTRA 32
static int redzone = -1;                              /* -1: not decided yet */
static void *(*real_malloc)(size_t), *(*real_calloc)(size_t, size_t), *(*real_realloc)(void *, size_t), *(*real_memalign)(size_t, size_t);
static char boot[1 << 16];                            /* what dlsym itself allocates while the real functions are being looked up */
static size_t boot_used;
static __thread int rz_inside __attribute__((tls_model("initial-exec")));

static void rz_init(void) {
    if (redzone >= 0) return;
    rz_inside++;
    real_malloc = (void *(*)(size_t))dlsym(RTLD_NEXT, "malloc");
    real_calloc = (void *(*)(size_t, size_t))dlsym(RTLD_NEXT, "calloc");
    real_realloc = (void *(*)(void *, size_t))dlsym(RTLD_NEXT, "realloc");
    real_memalign = (void *(*)(size_t, size_t))dlsym(RTLD_NEXT, "memalign");
    const char *s = getenv("HEAPGUARD_REDZONE");
    redzone = (s && *s == '1') ? 1 : 0;
    rz_inside--;
}
static int in_boot(const void *p) { return (const char *)p >= boot && (const char *)p < boot + sizeof boot; }
static void *boot_alloc(size_t n) {
    size_t at = (boot_used + 15) & ~(size_t)15;
    if (at + n > sizeof boot) return NULL;
    boot_used = at + n;
    return boot + at;
}
static void rz_arm(void *p, size_t n, void *site) {
    if (!p) return;
    const size_t usable = malloc_usable_size(p);
    if (usable < n + RZ_EXTRA) return;
    uint64_t c = RZ_CANARY, t[3] = {(uint64_t)n ^ RZ_MAGIC1, (uint64_t)(uintptr_t)site, RZ_MAGIC2};
    memcpy((char *)p + n, &c, 8);
    memcpy((char *)p + usable - 24, t, 24);
    __atomic_add_fetch(&n_zoned, 1, __ATOMIC_RELAXED);
}
/* 1 = the block carries a trailer (and *n is the size asked for); reports a damaged canary */
static int rz_check(void *p, size_t *n, const char *when) {
    const size_t usable = malloc_usable_size(p);
    if (usable < RZ_EXTRA) return 0;
    uint64_t t[3], c;
    memcpy(t, (char *)p + usable - 24, 24);
    if (t[2] != RZ_MAGIC2) return 0;
    const uint64_t asked = t[0] ^ RZ_MAGIC1;
    if (asked + RZ_EXTRA > usable) return 0;
    *n = (size_t)asked;
    memcpy(&c, (char *)p + asked, 8);
    if (c != RZ_CANARY) {
        __atomic_add_fetch(&n_overruns, 1, __ATOMIC_RELAXED);
        FILE *outs[2] = {stderr, logf};
        for (int k = 0; k < 2; k++) {
            FILE *f = outs[k];
            if (!f) continue;
            fprintf(f, "[heapguard] WRITE PAST THE END OF A LIVE BLOCK (%s): block %p of %zu bytes; the 8 bytes behind it read", when, p, (size_t)asked);
            for (int i = 0; i < 8; i++) fprintf(f, " %02x", ((unsigned char *)p)[asked + i]);
            fprintf(f, " (canary c5 x 8)\n    last bytes of the block as text: \"");
            for (size_t i = asked > 24 ? asked - 24 : 0; i < asked; i++) { unsigned char b = ((unsigned char *)p)[i]; fputc(b >= 32 && b < 127 ? b : '.', f); }
            fprintf(f, "\"\n    allocated from:\n");
            describe_frame(f, (void *)(uintptr_t)t[1]);
            fflush(f);
        }
    }
    memset((char *)p + usable - 24, 0, 24);           /* the trailer dies with the block */
    return 1;
}
void *malloc(size_t n) {
    if (rz_inside) return boot_alloc(n);
    rz_init();
    if (!redzone) return real_malloc(n);
    void *p = real_malloc(n + RZ_EXTRA);
    rz_arm(p, n, __builtin_return_address(0));
    return p;
}
void *calloc(size_t a, size_t b) {
    if (rz_inside) { void *p = boot_alloc(a * b); if (p) memset(p, 0, a * b); return p; }
    rz_init();
    if (!redzone) return real_calloc(a, b);
    if (b && a > (SIZE_MAX - RZ_EXTRA) / b) return NULL;
    void *p = real_calloc(1, a * b + RZ_EXTRA);
    rz_arm(p, a * b, __builtin_return_address(0));
    return p;
}
void *realloc(void *old, size_t n) {
    if (rz_inside) return boot_alloc(n);
    rz_init();
    if (in_boot(old)) { void *p = malloc(n); if (p && old) memcpy(p, old, n < 256 ? n : 256); return p; }
    if (!redzone) return real_realloc(old, n);
    size_t was = 0;
    if (old) (void)rz_check(old, &was, "realloc");
    void *p = real_realloc(old, n + RZ_EXTRA);
    rz_arm(p, n, __builtin_return_address(0));
    return p;
}
void *memalign(size_t al, size_t n) {
    rz_init();
    if (!redzone || rz_inside) return real_memalign(al, n);
    void *p = real_memalign(al, n + RZ_EXTRA);
    rz_arm(p, n, __builtin_return_address(0));
    return p;
}
void *aligned_alloc(size_t al, size_t n) { return memalign(al, n); }
int posix_memalign(void **out, size_t al, size_t n) {
    void *p = memalign(al, n);
    if (!p) return 12;
    *out = p;
    return 0;
}
static void guarded_free(void *p, void *site) {
    if (!p || in_boot(p)) return;
    if (redzone == 1) { size_t n; (void)rz_check(p, &n, "free"); }
    park(p, site);
}

/* ---- pinned host memory (HIP) ----------------------------------------------------------------------------------------------- */
typedef struct { void *p; size_t size; double t; } pinned;
#define NPIN 4096
static pinned pin_live[NPIN];                          /* ranges handed out by hipHostMalloc / registered by hipHostRegister */
static pinned *pin_ring;
static size_t pin_cap = 64, pin_head, pin_count;
static pthread_mutex_t pin_mu = PTHREAD_MUTEX_INITIALIZER;
static int (*real_hipHostMalloc)(void **, size_t, unsigned), (*real_hipHostFree)(void *), (*real_hipHostRegister)(void *, size_t, unsigned), (*real_hipHostUnregister)(void *);
static int (*real_hipDeviceSynchronize)(void);
static int pin_quarantine = 1;                        /* HEAPGUARD_PINNED=0: log the ranges only */
/* the HIP runtime is usually brought in by a dlopen with RTLD_LOCAL (Python extension modules): RTLD_NEXT does not see it */
static void *hip_sym(const char *name) {
    void *f = dlsym(RTLD_NEXT, name);
    static const char *libs[] = {"libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", NULL};
    for (int i = 0; !f && libs[i]; i++) {
        void *h = dlopen(libs[i], RTLD_LAZY | RTLD_NOLOAD);
        if (h) f = dlsym(h, name);
    }
    if (!f) { fprintf(stderr, "[heapguard] cannot find the runtime's %s\n", name); abort(); }
    return f;
}
static void pin_init(void) {
    if (real_hipHostFree) return;
    real_hipDeviceSynchronize = (int (*)(void))hip_sym("hipDeviceSynchronize");
    { const char *q = getenv("HEAPGUARD_PINNED"); if (q && *q == '0') pin_quarantine = 0; }
    real_hipHostMalloc = (int (*)(void **, size_t, unsigned))hip_sym("hipHostMalloc");
    real_hipHostRegister = (int (*)(void *, size_t, unsigned))hip_sym("hipHostRegister");
    real_hipHostUnregister = (int (*)(void *))hip_sym("hipHostUnregister");
    const char *s = getenv("HEAPGUARD_PINNED_RING");
    if (s) pin_cap = (size_t)strtoull(s, NULL, 10);
    if (pin_cap < 1) pin_cap = 1;
    pin_ring = (pinned *)calloc(pin_cap, sizeof(pinned));
    real_hipHostFree = (int (*)(void *))hip_sym("hipHostFree");
}
static void pin_log(const char *what, void *p, size_t n) {
    if (logf) { fprintf(logf, "[heapguard] %.6f %s %p .. %p (%zu bytes)\n", now(), what, p, (char *)p + n, n); fflush(logf); }
}
static void pin_release(pinned *e, const char *when) {
    const unsigned char *b = (const unsigned char *)e->p;
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < e->size; i++) if (b[i] != FILL) { if (!bad) first = i; bad++; }
    if (bad) {
        n_pin_violations++;
        FILE *outs[2] = {stderr, logf};
        for (int k = 0; k < 2; k++) if (outs[k]) {
            fprintf(outs[k], "[heapguard] WRITE INTO RELEASED PINNED MEMORY (%s): range %p of %zu bytes, released %.6f s ago; %zu byte(s) changed from offset %zu:", when, e->p, e->size, now() - e->t, bad, first);
            for (size_t i = first; i < e->size && i < first + 32; i++) fprintf(outs[k], " %02x", b[i]);
            fprintf(outs[k], "\n"); fflush(outs[k]);
        }
    }
    real_hipHostFree(e->p);
    e->p = NULL;
}
int hipHostMalloc(void **out, size_t n, unsigned flags) {
    if (!ready) { static pthread_once_t once = PTHREAD_ONCE_INIT; pthread_once(&once, init); }
    pin_init();
    const int rc = real_hipHostMalloc(out, n, flags);
    if (rc == 0 && out && *out) {
        pthread_mutex_lock(&pin_mu);
        n_pin_allocs++;
        for (int i = 0; i < NPIN; i++) if (!pin_live[i].p) { pin_live[i].p = *out; pin_live[i].size = n; pin_live[i].t = now(); break; }
        pthread_mutex_unlock(&pin_mu);
        pin_log("hipHostMalloc", *out, n);
    }
    return rc;
}
int hipHostFree(void *p) {
    pin_init();
    if (!p || disabled) return real_hipHostFree(p);
    size_t n = 0;
    pthread_mutex_lock(&pin_mu);
    for (int i = 0; i < NPIN; i++) if (pin_live[i].p == p) { n = pin_live[i].size; pin_live[i].p = NULL; break; }
    pthread_mutex_unlock(&pin_mu);
    pin_log("hipHostFree", p, n);
    if (!n || !pin_quarantine) return real_hipHostFree(p);   /* not one of ours (allocated before the preload saw it) */
    if (real_hipDeviceSynchronize) (void)real_hipDeviceSynchronize();      /* what the real call does first: queued work may still read or write the range */
    memset(p, FILL, n);                                 /* the runtime believes the range is still in use: it stays mapped and pinned */
    pthread_mutex_lock(&pin_mu);
    if (pin_count == pin_cap) { pinned *old = &pin_ring[pin_head]; pin_head = (pin_head + 1) % pin_cap; pin_count--; if (old->p) pin_release(old, "leaving the ring"); }
    pinned *e = &pin_ring[(pin_head + pin_count) % pin_cap];
    e->p = p; e->size = n; e->t = now();
    pin_count++;
    pthread_mutex_unlock(&pin_mu);
    return 0;
}
int hipHostRegister(void *p, size_t n, unsigned flags) {
    pin_init();
    const int rc = real_hipHostRegister(p, n, flags);
    if (rc == 0) pin_log("hipHostRegister", p, n);
    return rc;
}
int hipHostUnregister(void *p) {
    pin_init();
    pin_log("hipHostUnregister", p, 0);
    return real_hipHostUnregister(p);
}
static void pin_sweep_at_exit(void) {
    if (!pin_ring) return;
    for (size_t k = 0; k < pin_count; k++) {
        pinned *e = &pin_ring[(pin_head + k) % pin_cap];
        if (!e->p) continue;
        const unsigned char *b = (const unsigned char *)e->p;
        for (size_t i = 0; i < e->size; i++) if (b[i] != FILL) { n_pin_violations++; if (logf) fprintf(logf, "[heapguard] WRITE INTO RELEASED PINNED MEMORY (at exit): range %p of %zu bytes, offset %zu holds %02x\n", e->p, e->size, i, b[i]); break; }
    }
}

void free(void *p) { guarded_free(p, __builtin_return_address(0)); }
/* C++ deletes (libstdc++'s go to free(): interposed here so that the recorded site is the caller of `delete`, not libstdc++) */
void _ZdlPv(void *p) { guarded_free(p, __builtin_return_address(0)); }
void _ZdaPv(void *p) { guarded_free(p, __builtin_return_address(0)); }
void _ZdlPvm(void *p, unsigned long n) { (void)n; guarded_free(p, __builtin_return_address(0)); }
void _ZdaPvm(void *p, unsigned long n) { (void)n; guarded_free(p, __builtin_return_address(0)); }
void _ZdlPvSt11align_val_t(void *p, unsigned long a) { (void)a; guarded_free(p, __builtin_return_address(0)); }
void _ZdlPvmSt11align_val_t(void *p, unsigned long n, unsigned long a) { (void)n; (void)a; guarded_free(p, __builtin_return_address(0)); }
void _ZdaPvSt11align_val_t(void *p, unsigned long a) { (void)a; guarded_free(p, __builtin_return_address(0)); }
void _ZdaPvmSt11align_val_t(void *p, unsigned long n, unsigned long a) { (void)n; (void)a; guarded_free(p, __builtin_return_address(0)); }

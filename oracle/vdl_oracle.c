/*
 * vdl_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * A scalar, single-threaded, op-at-a-time CPU interpreter for the textual VDL
 * (Voodoo vector-operator dataflow) that orm011/mplan2vdl prints, plus
 * independent fused scalar "SQL-semantics" evaluators for TPC-H Q6 / Q1 and the
 * counter-based synthetic column generator.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (mplan2vdl_amd/lib/libvdl.so) never links or calls it.
 *
 * PARITY STATUS: the reference repository contains no executor (its VDL text is
 * POSTed to an external Voodoo server, /root/reference/eval_query.sh:21-24), so
 * result parity with the original Voodoo backend is UNPINNED.  What pins this
 * oracle instead:
 *   (1) grammar / operand order  : /root/reference/src/Vdl.hs:410-477 and the 12
 *       golden VDL lines in /root/reference/README.md:40-52 (tests/test_fixtures.py);
 *   (2) values                   : the fused SQL-semantics loops below (orc_sql_q6,
 *       orc_sql_q1), written from the SQL text in the plan headers
 *       (/root/reference/tests/tpch10noorder/06.sql.mplan:1-9, 01.sql.mplan:1-18),
 *       must agree bit-for-bit with this interpreter running the VDL programs;
 *   (3) constants                : /root/reference/README.md:44,48 (dates);
 *   (4) independent lowerings    : the same SQL compiled three ways by the front end restatement (FK join-index
 *       gathers, --crossproduct, the VLite format's Semisort grouping) must give the same answers through this
 *       interpreter (tests/test_tpch_plans.py), Q3 must equal a numpy evaluation of its SQL (tests/helpers.py:sql_q3),
 *       and Like must equal a regular-expression statement of SQL LIKE (tests/test_oracle.py).
 *
 * Vector model (normative for this repo, see DESIGN.md "Semantics"):
 *   a vector has n slots, each an int64 value or EPS (empty).  Filters never
 *   shrink vectors; only MaterializeCompact drops EPS.  RangeV inherits EPS from
 *   its size-reference vector (needed so that `count(*)` lowered as
 *   FoldSum(zeros_ refv, ones_ refv), /root/reference/src/Vlite.hs:636-639,
 *   1044-1046,982-983, counts only selected rows).  Fold runs skip EPS control
 *   slots.
 *
 * Every operator function cites the reference lines it restates.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <time.h>
#include <ctype.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_FIELD 192
#define ORC_MATERIALIZE_LIMIT ((int64_t)1 << 33) /* refuse to materialise ranges longer than this */

typedef struct {
    int      defined;
    int64_t  n;
    int64_t *val;            /* NULL for a virtual range */
    uint8_t *ok;             /* NULL = all slots hold a value */
    int      is_range;       /* RangeC kept virtual: val_i = from + i*step */
    int64_t  from, step;
    int      owns;
    char     field[ORC_MAX_FIELD];
} ovec;

typedef struct {
    char    name[ORC_MAX_FIELD];
    const void *data;
    int     elem_bytes;
    int64_t n;
} ocol;

typedef struct {
    char     name[ORC_MAX_FIELD];
    char     tmp[32];
    int64_t *vals;
    int64_t  n;
} oout;

typedef struct orc_ctx {
    ocol   *cols; int ncols, capcols;
    ovec   *vecs; int nvecs;
    oout   *outs; int nouts, capouts;
    char    err[512];
    double  last_seconds;
    int64_t ops_executed;
    int     keep_vectors;     /* orc_keep_vectors: intermediates survive orc_run (statement-by-statement comparison in tests) */
} orc_ctx;

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int fail(orc_ctx *c, const char *fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(c->err, sizeof c->err, fmt, ap);
    va_end(ap);
    return -1;
}

/* ------------------------------------------------------------------ context */

orc_ctx *orc_open(void) { return (orc_ctx *)calloc(1, sizeof(orc_ctx)); }

static void free_vec(ovec *v) {
    if (v->owns) { free(v->val); free(v->ok); }
    memset(v, 0, sizeof *v);
}

static void clear_run_state(orc_ctx *c) {
    for (int i = 0; i < c->nvecs; i++) free_vec(&c->vecs[i]);
    free(c->vecs); c->vecs = NULL; c->nvecs = 0;
    for (int i = 0; i < c->nouts; i++) free(c->outs[i].vals);
    c->nouts = 0;
}

void orc_close(orc_ctx *c) {
    if (!c) return;
    clear_run_state(c);
    free(c->outs); free(c->cols); free(c);
}

const char *orc_last_error(const orc_ctx *c) { return c->err; }
double orc_last_run_seconds(const orc_ctx *c) { return c->last_seconds; }
int64_t orc_last_ops(const orc_ctx *c) { return c->ops_executed; }

/* Borrowed host pointer; little-endian signed integers of elem_bytes in {1,2,4,8}
 * (storage widths: /root/reference/tests/tpch10noorder/storage.csv:188-208). */
int orc_add_column(orc_ctx *c, const char *name, const void *data, int elem_bytes, int64_t n) {
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8)
        return fail(c, "column %s: unsupported width %d", name, elem_bytes);
    for (int i = 0; i < c->ncols; i++)
        if (!strcmp(c->cols[i].name, name)) {
            c->cols[i].data = data; c->cols[i].elem_bytes = elem_bytes; c->cols[i].n = n;
            return 0;
        }
    if (c->ncols == c->capcols) {
        c->capcols = c->capcols ? 2 * c->capcols : 16;
        c->cols = (ocol *)realloc(c->cols, sizeof(ocol) * (size_t)c->capcols);
    }
    ocol *k = &c->cols[c->ncols++];
    snprintf(k->name, sizeof k->name, "%s", name);
    k->data = data; k->elem_bytes = elem_bytes; k->n = n;
    return 0;
}

/* ------------------------------------------------------------------ helpers */

static inline int slot_ok(const ovec *v, int64_t i) { return v->ok ? v->ok[i] : 1; }
static inline int64_t slot_val(const ovec *v, int64_t i) {
    return v->is_range ? (int64_t)((uint64_t)v->from + (uint64_t)i * (uint64_t)v->step) : v->val[i];
}

static int alloc_vec(orc_ctx *c, ovec *v, int64_t n, int with_ok) {
    memset(v, 0, sizeof *v);
    v->defined = 1; v->n = n; v->owns = 1;
    v->val = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    if (!v->val) return fail(c, "out of memory (%lld slots)", (long long)n);
    if (with_ok) {
        v->ok = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
        if (!v->ok) return fail(c, "out of memory (%lld slots)", (long long)n);
    }
    strcpy(v->field, "val");
    return 0;
}

/* wrap-around int64 arithmetic without UB */
static inline int64_t w_add(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
static inline int64_t w_sub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
static inline int64_t w_mul(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }
static inline int64_t w_div(int64_t a, int64_t b) {           /* C truncation; x/0 := 0 */
    if (b == 0) return 0;
    if (b == -1) return (int64_t)(0 - (uint64_t)a);
    return a / b;
}
static inline int64_t w_mod(int64_t a, int64_t b) {           /* C remainder; x%0 := 0 */
    if (b == 0 || b == -1) return 0;
    return a % b;
}
/* BitShift: sign of the amount encodes direction (/root/reference/src/Vlite.hs:205-208,
 * 453-455): b >= 0 arithmetic right shift, b < 0 left shift by -b. */
static inline int64_t w_shift(int64_t a, int64_t b) {
    if (b >= 0) { if (b > 63) b = 63; return a >> b; }
    if (b <= -64) return 0;
    return (int64_t)((uint64_t)a << (unsigned)(-b));
}

enum { OP_LAND, OP_LOR, OP_BAND, OP_BOR, OP_SHIFT, OP_EQ, OP_ADD, OP_SUB, OP_GT, OP_MUL, OP_DIV, OP_MOD, OP_NBIN };
static const char *BIN_NAMES[OP_NBIN] = { "LogicalAnd", "LogicalOr", "BitwiseAnd", "BitwiseOr", "BitShift",
    "Equals", "Add", "Subtract", "Greater", "Multiply", "Divide", "Modulo" };

static inline int64_t apply_bin(int op, int64_t a, int64_t b) {
    switch (op) {
    case OP_LAND:  return (a != 0) && (b != 0);
    case OP_LOR:   return (a != 0) || (b != 0);
    case OP_BAND:  return a & b;
    case OP_BOR:   return a | b;
    case OP_SHIFT: return w_shift(a, b);
    case OP_EQ:    return a == b;
    case OP_ADD:   return w_add(a, b);
    case OP_SUB:   return w_sub(a, b);
    case OP_GT:    return a > b;
    case OP_MUL:   return w_mul(a, b);
    case OP_DIV:   return w_div(a, b);
    default:       return w_mod(a, b);
    }
}

/* ------------------------------------------------------------------ operators */

/* Load  (/root/reference/src/Vdl.hs:161-168,419-420): column -> vector whose
 * struct field is the key path minus its first component. */
static int op_load(orc_ctx *c, ovec *out, const char *name) {
    const ocol *k = NULL;
    for (int i = 0; i < c->ncols; i++) if (!strcmp(c->cols[i].name, name)) k = &c->cols[i];
    if (!k) return fail(c, "Load: unknown column '%s'", name);
    if (alloc_vec(c, out, k->n, 0)) return -1;
    for (int64_t i = 0; i < k->n; i++) {
        switch (k->elem_bytes) {
        case 1: out->val[i] = ((const int8_t  *)k->data)[i]; break;
        case 2: out->val[i] = ((const int16_t *)k->data)[i]; break;
        case 4: out->val[i] = ((const int32_t *)k->data)[i]; break;
        default: out->val[i] = ((const int64_t *)k->data)[i]; break;
        }
    }
    const char *dot = strchr(name, '.');
    snprintf(out->field, sizeof out->field, "%s", dot ? dot + 1 : name);
    return 0;
}

/* Project out,v,in (/root/reference/src/Vdl.hs:34,422-423): full rename of one field. */
static int op_project(orc_ctx *c, ovec *out, const ovec *v, const char *outf, const char *inf) {
    if (strcmp(v->field, inf)) return fail(c, "Project: operand has field '%s', not '%s'", v->field, inf);
    *out = *v; out->owns = 0;                 /* alias: same slots */
    snprintf(out->field, sizeof out->field, "%s", outf);
    return 0;
}

/* RangeV from,v,step (/root/reference/src/Vdl.hs:428-431; Vlite.hs:299-306): len(v)
 * slots, value from + i*step, EPS where v is EPS (see header). */
static int op_rangev(orc_ctx *c, ovec *out, int64_t from, const ovec *v, int64_t step) {
    if (alloc_vec(c, out, v->n, v->ok != NULL)) return -1;
    for (int64_t i = 0; i < v->n; i++) {
        out->val[i] = (int64_t)((uint64_t)from + (uint64_t)i * (uint64_t)step);
        if (out->ok) out->ok[i] = v->ok[i];
    }
    return 0;
}

/* RangeC from,count,step (/root/reference/src/Vdl.hs:433-434; Vlite.hs:308-314).
 * Kept virtual: Q3's pivots are RangeC 0 2^38 1. */
static int op_rangec(orc_ctx *c, ovec *out, int64_t from, int64_t count, int64_t step) {
    if (count < 0) return fail(c, "RangeC: negative count");
    memset(out, 0, sizeof *out);
    out->defined = 1; out->n = count; out->is_range = 1; out->from = from; out->step = step;
    strcpy(out->field, "val");
    return 0;
}

/* element-wise binary (/root/reference/src/Vdl.hs:110-122,436-439; bounds Vlite.hs:417-458):
 * slot-wise, EPS if either side EPS. */
static int op_binary(orc_ctx *c, ovec *out, int op, const ovec *a, const ovec *b) {
    if (a->n != b->n) return fail(c, "%s: operand lengths differ (%lld vs %lld)", BIN_NAMES[op], (long long)a->n, (long long)b->n);
    int with_ok = (a->ok || b->ok);
    if (alloc_vec(c, out, a->n, with_ok)) return -1;
    for (int64_t i = 0; i < a->n; i++) {
        int ok = slot_ok(a, i) && slot_ok(b, i);
        if (with_ok) out->ok[i] = (uint8_t)ok;
        out->val[i] = ok ? apply_bin(op, slot_val(a, i), slot_val(b, i)) : 0;
    }
    return 0;
}

/* Run segmentation shared by all folds (/root/reference/src/Vlite.hs:337-356 and the
 * lowering at :1048-1064): a run is a maximal stretch of control slots holding equal
 * values, EPS control slots being skipped.  Calls cb(first_slot, members...) lazily:
 * we walk slots in order and keep the current run's first slot. */
enum { F_SEL, F_SUM, F_MIN, F_MAX, F_CHOOSE, F_COUNT };

static int op_fold(orc_ctx *c, ovec *out, int kind, const ovec *ctl, const ovec *d) {
    if (ctl->n != d->n) return fail(c, "Fold: control and data lengths differ (%lld vs %lld)", (long long)ctl->n, (long long)d->n);
    int64_t n = d->n;
    if (alloc_vec(c, out, n, 1)) return -1;
    memset(out->ok, 0, (size_t)n);
    memset(out->val, 0, sizeof(int64_t) * (size_t)n);
    int have_run = 0; int64_t run_key = 0, run_first = -1;
    int64_t acc = 0; int acc_ok = 0;
    int64_t sel_write = -1;            /* FoldSelect: next member slot to write in this run */
    /* FoldSelect needs the member slots of the run in order: with EPS-skipping the
     * members are exactly the non-EPS control slots, so we advance sel_write over them. */
    for (int64_t i = 0; i < n; i++) {
        if (!slot_ok(ctl, i)) continue;
        int64_t k = slot_val(ctl, i);
        if (!have_run || k != run_key) {
            if (have_run && kind != F_SEL && acc_ok) { out->val[run_first] = acc; out->ok[run_first] = 1; }
            /* The first run of a value fold starts at slot 0: EPS control slots ahead of it belong to it.  The compiler
             * relies on it -- the result of an ungrouped aggregate is a one-row relation (count = 1) that it broadcasts
             * with Gather(result, zeros_ other), i.e. reads at position 0 (/root/reference/src/Vlite.hs:693-712, `@@ zeros_`;
             * TPC-H Q11's HAVING threshold) -- while the aggregate's input is a FILTERED vector whose slot 0 is EPS
             * unless row 0 happens to pass the filter.  FoldSelect keeps the slot of the member itself. */
            run_first = (!have_run && kind != F_SEL) ? 0 : i;
            have_run = 1; run_key = k; acc = 0; acc_ok = 0; sel_write = i;
        }
        if (!slot_ok(d, i)) continue;
        int64_t x = slot_val(d, i);
        switch (kind) {
        case F_SEL:
            /* FoldSelect (/root/reference/src/Vdl.hs:124,263; Vlite.hs:331-335,727):
             * positions of non-zero data, packed at the start of the run. */
            if (x != 0) {
                while (!slot_ok(ctl, sel_write)) sel_write++;   /* next member slot */
                out->val[sel_write] = i; out->ok[sel_write] = 1; sel_write++;
            }
            break;
        case F_SUM:    acc = w_add(acc, x); acc_ok = 1; break;
        case F_MIN:    acc = (!acc_ok || x < acc) ? x : acc; acc_ok = 1; break;
        case F_MAX:    acc = (!acc_ok || x > acc) ? x : acc; acc_ok = 1; break;
        case F_CHOOSE: if (!acc_ok) { acc = x; acc_ok = 1; } break;  /* any element: the first */
        default:       acc = w_add(acc, 1); acc_ok = 1; break;
        }
    }
    if (have_run && kind != F_SEL && acc_ok) { out->val[run_first] = acc; out->ok[run_first] = 1; }
    return 0;
}

/* Gather src,pos (/root/reference/src/Vdl.hs:129,238,438; Vlite.hs:325-329): out_i = src[pos_i];
 * EPS if pos_i is EPS, out of range, or the source slot is EPS. */
static int op_gather(orc_ctx *c, ovec *out, const ovec *src, const ovec *pos) {
    if (alloc_vec(c, out, pos->n, 1)) return -1;
    for (int64_t i = 0; i < pos->n; i++) {
        int ok = slot_ok(pos, i);
        int64_t p = ok ? slot_val(pos, i) : 0;
        ok = ok && p >= 0 && p < src->n && slot_ok(src, p);
        out->ok[i] = (uint8_t)ok;
        out->val[i] = ok ? slot_val(src, p) : 0;
    }
    return 0;
}

/* Scatter src,fold,pos (/root/reference/src/Vdl.hs:38,239-242,441-442; Vlite.hs:316-320):
 * out[pos_i] = src_i; output length = len(fold operand); unwritten slots EPS.  Positions
 * are unique at every call site (Vlite.hs:1267, :508); on duplicates the later slot wins. */
static int op_scatter(orc_ctx *c, ovec *out, const ovec *src, const ovec *fold, const ovec *pos) {
    if (src->n != pos->n) return fail(c, "Scatter: source and position lengths differ");
    int64_t n = fold->n;
    if (alloc_vec(c, out, n, 1)) return -1;
    memset(out->ok, 0, (size_t)n);
    memset(out->val, 0, sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < src->n; i++) {
        if (!slot_ok(src, i) || !slot_ok(pos, i)) continue;
        int64_t p = slot_val(pos, i);
        if (p < 0 || p >= n) continue;
        out->val[p] = slot_val(src, i); out->ok[p] = 1;
    }
    return 0;
}

/* Partition data,pivots (/root/reference/src/Vdl.hs:130,266-269; Vlite.hs:358-366,508,
 * 1082-1098): bucket_i = index of the first pivot >= data_i (pivots ascending; with the
 * emitted pivots = RangeC min cnt 1 this is data_i - min); the result is the stable
 * counting-sort destination of every non-EPS slot -- a permutation of 0..m-1 (it is
 * scattered by and declared Unique, Vlite.hs:508,1058-1059).  EPS in -> EPS out. */
static int64_t bucket_of(const ovec *piv, int64_t x) {
    if (piv->is_range && piv->step == 1) {
        if (x <= piv->from) return 0;
        int64_t b = w_sub(x, piv->from);
        return b < piv->n ? b : piv->n;          /* beyond the last pivot: overflow bucket */
    }
    int64_t lo = 0, hi = piv->n;                 /* lower_bound over ascending pivots */
    while (lo < hi) { int64_t mid = lo + (hi - lo) / 2; if (slot_val(piv, mid) < x) lo = mid + 1; else hi = mid; }
    return lo;
}

typedef struct { int64_t bucket, idx; } bpair;
static int cmp_bpair(const void *a, const void *b) {
    const bpair *x = (const bpair *)a, *y = (const bpair *)b;
    if (x->bucket != y->bucket) return x->bucket < y->bucket ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

static int op_partition(orc_ctx *c, ovec *out, const ovec *data, const ovec *piv) {
    int64_t n = data->n;
    if (piv->ok) return fail(c, "Partition: pivots with EPS slots are not supported");
    if (alloc_vec(c, out, n, 1)) return -1;
    memset(out->val, 0, sizeof(int64_t) * (size_t)n);
    int64_t nb = piv->n + 1;
    if (nb <= ((int64_t)1 << 24)) {              /* dense domain: counting sort */
        int64_t *cnt = (int64_t *)calloc((size_t)nb + 1, sizeof(int64_t));
        if (!cnt) return fail(c, "out of memory");
        for (int64_t i = 0; i < n; i++) { out->ok[i] = (uint8_t)slot_ok(data, i); if (out->ok[i]) cnt[bucket_of(piv, slot_val(data, i)) + 1]++; }
        for (int64_t b = 0; b < nb; b++) cnt[b + 1] += cnt[b];
        for (int64_t i = 0; i < n; i++) if (out->ok[i]) out->val[i] = cnt[bucket_of(piv, slot_val(data, i))]++;
        free(cnt);
    } else {                                     /* sparse domain: stable sort of (bucket, slot) */
        int64_t m = 0;
        bpair *p = (bpair *)malloc(sizeof(bpair) * (size_t)(n > 0 ? n : 1));
        if (!p) return fail(c, "out of memory");
        for (int64_t i = 0; i < n; i++) { out->ok[i] = (uint8_t)slot_ok(data, i); if (out->ok[i]) { p[m].bucket = bucket_of(piv, slot_val(data, i)); p[m].idx = i; m++; } }
        qsort(p, (size_t)m, sizeof(bpair), cmp_bpair);
        for (int64_t r = 0; r < m; r++) out->val[p[r].idx] = r;
        free(p);
    }
    return 0;
}

/* MaterializeCompact v (/root/reference/src/Vdl.hs:41,271-292,452-453): a query output;
 * drops EPS, keeps slot order; named by the operand's field (resolve.py:55-78). */
/* Like data dict pattern (/root/reference/src/Vdl.hs:39,244-247,444-447; Vlite.hs:113,296,1010-1014;
 * Mplan.hs:398-417,528-545).  `data` holds byte offsets into the string heap `dict` (the column's
 * "<table>.<col>.heap" vector, one byte per slot; MonetDB stores var-sized strings this way and the
 * codes of dictionary.csv are such offsets); the string runs to the next 0 byte or the end of the heap.
 * SQL LIKE without escape character (Mplan.hs:541-542 requires the empty escape): '%' = any run of bytes,
 * '_' = any one byte, everything else literal, case-sensitive.  Result 0/1 (Vlite.hs:296 bounds (0,1));
 * EPS where data is EPS; an offset outside the heap matches nothing. */
static int like_match(const ovec *heap, int64_t off, const char *pat) {
    if (off < 0 || off >= heap->n) return 0;
    const int64_t plen = (int64_t)strlen(pat);
    int64_t si = off, pi = 0, star = -1, mark = 0;
    for (;;) {
        const int ch = (si < heap->n && slot_ok(heap, si)) ? (int)(slot_val(heap, si) & 0xff) : 0;
        if (!ch) break;
        if (pi < plen && pat[pi] == '%') { star = pi++; mark = si; }
        else if (pi < plen && (pat[pi] == '_' || (unsigned char)pat[pi] == ch)) { si++; pi++; }
        else if (star >= 0) { pi = star + 1; si = ++mark; }
        else return 0;
    }
    while (pi < plen && pat[pi] == '%') pi++;
    return pi == plen;
}

static int op_like(orc_ctx *c, ovec *out, const ovec *data, const ovec *heap, const char *pat) {
    if (alloc_vec(c, out, data->n, data->ok != NULL)) return -1;
    for (int64_t i = 0; i < data->n; i++) {
        if (!slot_ok(data, i)) { out->ok[i] = 0; out->val[i] = 0; continue; }
        if (out->ok) out->ok[i] = 1;
        out->val[i] = like_match(heap, slot_val(data, i), pat);
    }
    return 0;
}

/* Semisort data (/root/reference/src/Vdl.hs:42,205-207,377; Vlite.hs:109,322,1061-1064; VLite format only):
 * a gather mask that brings equal values together -- "gmask = Semisort gkeyvec; fgroups = gkeyvec @@ gmask;
 * fdata = gdata @@ gmask; Fold fgroups fdata".  Normative reading: the slots of the non-EPS values in stable
 * ascending value order, packed at the front; the remaining slots EPS (so the gathered vectors are sorted
 * prefixes and the folds see one run per distinct value, in value order like the Partition lowering). */
static int op_semisort(orc_ctx *c, ovec *out, const ovec *d) {
    if (alloc_vec(c, out, d->n, 1)) return -1;
    bpair *tmp = (bpair *)malloc(sizeof(bpair) * (size_t)(d->n > 0 ? d->n : 1));
    if (!tmp) return fail(c, "out of memory");
    int64_t m = 0;
    for (int64_t i = 0; i < d->n; i++) if (slot_ok(d, i)) { tmp[m].bucket = slot_val(d, i); tmp[m].idx = i; m++; }
    qsort(tmp, (size_t)m, sizeof(bpair), cmp_bpair);
    for (int64_t k = 0; k < d->n; k++) { out->ok[k] = k < m; out->val[k] = k < m ? tmp[k].idx : 0; }
    free(tmp);
    return 0;
}

static int op_materialize(orc_ctx *c, int id, const ovec *v) {
    if (c->nouts == c->capouts) {
        c->capouts = c->capouts ? 2 * c->capouts : 8;
        c->outs = (oout *)realloc(c->outs, sizeof(oout) * (size_t)c->capouts);
    }
    oout *o = &c->outs[c->nouts];
    memset(o, 0, sizeof *o);
    if (v->is_range && v->n > ORC_MATERIALIZE_LIMIT) return fail(c, "MaterializeCompact: range too long");
    int64_t m = 0;
    for (int64_t i = 0; i < v->n; i++) m += slot_ok(v, i);
    o->vals = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m > 0 ? m : 1));
    if (!o->vals) return fail(c, "out of memory");
    int64_t w = 0;
    for (int64_t i = 0; i < v->n; i++) if (slot_ok(v, i)) o->vals[w++] = slot_val(v, i);
    o->n = m;
    snprintf(o->name, sizeof o->name, "%s", v->field);
    snprintf(o->tmp, sizeof o->tmp, "tmp%d", id);
    c->nouts++;
    return 0;
}

/* ------------------------------------------------------------------ parser */
/* Line grammar: "<id>,<Op>,<fields...>"; operands print as "Id <n>"
 * (/root/reference/src/Vdl.hs:97-99,410-453,476-477).  Anything from " ;;" on is
 * metadata (Vdl.hs:463-466) and ignored, as eval_query.sh:20 strips it. */

#define MAXF 8
static int split_fields(char *line, char **f, int maxf, int keep_tail_from) {
    int nf = 0; char *p = line;
    while (nf < maxf) {
        f[nf++] = p;
        if (nf == keep_tail_from) break;          /* last field swallows the rest (Like pattern) */
        char *q = strchr(p, ',');
        if (!q) break;
        *q = 0; p = q + 1;
    }
    return nf;
}

static int parse_int(const char *s, int64_t *out) {
    while (*s == ' ') s++;
    char *end;
    long long v = strtoll(s, &end, 10);
    while (*end == ' ') end++;
    if (end == s || *end) return -1;
    *out = (int64_t)v; return 0;
}

static int parse_ref(const char *s, int *out) {
    while (*s == ' ') s++;
    if (strncmp(s, "Id ", 3)) return -1;
    int64_t v; if (parse_int(s + 3, &v) || v <= 0 || v > (1 << 24)) return -1;
    *out = (int)v; return 0;
}

static const ovec *get_vec(orc_ctx *c, int id) {
    if (id <= 0 || id >= c->nvecs || !c->vecs[id].defined) { fail(c, "reference to undefined vector Id %d", id); return NULL; }
    return &c->vecs[id];
}

static int need_field(orc_ctx *c, const ovec *v, const char *f, const char *opname) {
    if (strcmp(v->field, f)) return fail(c, "%s: operand field is '%s', expected '%s'", opname, v->field, f);
    return 0;
}

static int exec_line(orc_ctx *c, char *line, int lineno) {
    char *cut = strstr(line, ";;"); if (cut) *cut = 0;
    size_t L = strlen(line);
    while (L && isspace((unsigned char)line[L - 1])) line[--L] = 0;
    char *s = line; while (*s && isspace((unsigned char)*s)) s++;
    if (!*s) return 0;
    char *f[MAXF + 1]; memset(f, 0, sizeof f);
    int nf = split_fields(s, f, MAXF, MAXF);        /* an 8th field (Like's pattern) keeps its commas */
    int64_t id64;
    if (nf < 2 || parse_int(f[0], &id64) || id64 <= 0 || id64 > (1 << 24)) return fail(c, "line %d: bad id", lineno);
    int id = (int)id64;
    if (id >= c->nvecs) {
        int newn = id + 64;
        c->vecs = (ovec *)realloc(c->vecs, sizeof(ovec) * (size_t)newn);
        memset(c->vecs + c->nvecs, 0, sizeof(ovec) * (size_t)(newn - c->nvecs));
        c->nvecs = newn;
    }
    if (c->vecs[id].defined) return fail(c, "line %d: Id %d defined twice", lineno, id);
    const char *op = f[1];
    ovec out; memset(&out, 0, sizeof out);
    int rc = -1, a, b, d;
    int64_t x, y;
    const ovec *va, *vb, *vd;
    c->ops_executed++;
#define NEED(k) if (nf != (k)) return fail(c, "line %d: %s expects %d fields, got %d", lineno, op, (k), nf)
    if (!strcmp(op, "Load")) {
        NEED(3); rc = op_load(c, &out, f[2]);
    } else if (!strcmp(op, "Project")) {
        NEED(5); if (parse_ref(f[3], &a) || !(va = get_vec(c, a))) return c->err[0] ? -1 : fail(c, "line %d: bad operand", lineno);
        rc = op_project(c, &out, va, f[2], f[4]);
    } else if (!strcmp(op, "RangeV")) {
        NEED(6); if (parse_int(f[3], &x) || parse_ref(f[4], &a) || parse_int(f[5], &y)) return fail(c, "line %d: bad RangeV", lineno);
        if (!(va = get_vec(c, a))) return -1;
        rc = op_rangev(c, &out, x, va, y);
    } else if (!strcmp(op, "RangeC")) {
        NEED(6); int64_t cnt; if (parse_int(f[3], &x) || parse_int(f[4], &cnt) || parse_int(f[5], &y)) return fail(c, "line %d: bad RangeC", lineno);
        rc = op_rangec(c, &out, x, cnt, y);
    } else if (!strcmp(op, "Gather")) {
        NEED(5); if (parse_ref(f[2], &a) || parse_ref(f[3], &b)) return fail(c, "line %d: bad Gather", lineno);
        if (!(va = get_vec(c, a)) || !(vb = get_vec(c, b))) return -1;
        if (need_field(c, vb, f[4], op)) return -1;
        rc = op_gather(c, &out, va, vb);
        if (!rc) snprintf(out.field, sizeof out.field, "%s", va->field);
    } else if (!strcmp(op, "Scatter")) {
        NEED(7); if (parse_ref(f[2], &a) || parse_ref(f[3], &b) || parse_ref(f[5], &d)) return fail(c, "line %d: bad Scatter", lineno);
        if (!(va = get_vec(c, a)) || !(vb = get_vec(c, b)) || !(vd = get_vec(c, d))) return -1;
        if (need_field(c, vb, f[4], op) || need_field(c, vd, f[6], op)) return -1;
        rc = op_scatter(c, &out, va, vb, vd);
        if (!rc) snprintf(out.field, sizeof out.field, "%s", va->field);
    } else if (!strcmp(op, "Shuffle")) {
        /* Shuffle v (/root/reference/src/Vdl.hs:40,449-450; Vlite.hs:294): value identity. */
        NEED(3); if (parse_ref(f[2], &a) || !(va = get_vec(c, a))) return c->err[0] ? -1 : fail(c, "line %d: bad operand", lineno);
        out = *va; out.owns = 0; rc = 0;
    } else if (!strcmp(op, "CrossProductOuter") || !strcmp(op, "CrossProductInner")) {
        /* <op>,Id left,Id right (Vdl.hs:412-416).  Vlite.hs:278-289: "0,1,2,3 X 0,1 = 0,0,1,1,2,2,3,3 (outer)
         * 0,1,0,1,0,1,0,1 (inner)": positions into left / right over len(left)*len(right) slots, never EPS. */
        NEED(4); if (parse_ref(f[2], &a) || parse_ref(f[3], &b)) return fail(c, "line %d: bad operands for %s", lineno, op);
        if (!(va = get_vec(c, a)) || !(vb = get_vec(c, b))) return -1;
        if (vb->n > 0 && va->n > ((int64_t)1 << 24) / vb->n) return fail(c, "line %d: cross product of %lld x %lld slots", lineno, (long long)va->n, (long long)vb->n);
        if (alloc_vec(c, &out, va->n * vb->n, 0)) return -1;
        for (int64_t i = 0; i < out.n; i++) out.val[i] = op[12] == 'I' ? i % vb->n : i / vb->n;
        rc = 0;
    } else if (!strcmp(op, "Like")) {
        /* Like,val,Id data,val,Id dict,val,<pattern>  (Vdl.hs:444-447) */
        NEED(8); if (parse_ref(f[3], &a) || parse_ref(f[5], &b)) return fail(c, "line %d: bad operands for Like", lineno);
        if (!(va = get_vec(c, a)) || !(vb = get_vec(c, b))) return -1;
        if (need_field(c, va, f[4], op) || need_field(c, vb, f[6], op)) return -1;
        rc = op_like(c, &out, va, vb, f[7]);
        if (!rc) snprintf(out.field, sizeof out.field, "%s", f[2]);
    } else if (!strcmp(op, "MaterializeCompact")) {
        NEED(3); if (parse_ref(f[2], &a) || !(va = get_vec(c, a))) return c->err[0] ? -1 : fail(c, "line %d: bad operand", lineno);
        if (op_materialize(c, id, va)) return -1;
        out = *va; out.owns = 0; rc = 0;
    } else {
        int bop = -1, fold = -1, part = 0;
        for (int k = 0; k < OP_NBIN; k++) if (!strcmp(op, BIN_NAMES[k])) bop = k;
        if      (!strcmp(op, "FoldSelect")) fold = F_SEL;
        else if (!strcmp(op, "FoldSum"))    fold = F_SUM;
        else if (!strcmp(op, "FoldMin"))    fold = F_MIN;
        else if (!strcmp(op, "FoldMax"))    fold = F_MAX;
        else if (!strcmp(op, "FoldChoose")) fold = F_CHOOSE;
        else if (!strcmp(op, "FoldCount"))  fold = F_COUNT;
        else if (!strcmp(op, "Partition"))  part = 1;
        if (bop < 0 && fold < 0 && !part) return fail(c, "line %d: unsupported operator '%s'", lineno, op);
        NEED(7);
        if (parse_ref(f[3], &a) || parse_ref(f[5], &b)) return fail(c, "line %d: bad operands for %s", lineno, op);
        if (!(va = get_vec(c, a)) || !(vb = get_vec(c, b))) return -1;
        if (need_field(c, va, f[4], op) || need_field(c, vb, f[6], op)) return -1;
        if (bop >= 0)      rc = op_binary(c, &out, bop, va, vb);
        else if (fold >= 0) rc = op_fold(c, &out, fold, va, vb);
        else               rc = op_partition(c, &out, va, vb);
        if (!rc) snprintf(out.field, sizeof out.field, "%s", f[2]);
    }
#undef NEED
    if (rc) return -1;
    out.defined = 1;
    c->vecs[id] = out;
    return 0;
}

/* ---- the VLite dialect ("lighter syntax (one value per vector)", /root/reference/src/MainFuns.hs:70) --------
 * Printed by toVList / printLine (Vdl.hs:370-408,455-475): no field names, operands in the same order,
 *   <id>,Load,<name> | <id>,Project,Id v | <id>,RangeV,<from>,Id v,<step> | <id>,RangeC,<from>,<count>,<step>
 *   <id>,<BinOp|Fold*|Partition|Gather>,Id a,Id b | <id>,Scatter,Id src,Id fold,Id pos | <id>,Semisort,Id v
 *   <id>,Shuffle,Id v | <id>,Like,Id data,Id dict,<pattern> | <id>,CrossProduct{Outer,Inner},Id l,Id r
 *   <id>,Output,Id v     or     <name>,Output,<display type>,Id v    (the id of a named output is not printed:
 *                                                                    it is the previous statement's id + 1)
 * A program is read as VLite when it has an Output statement. */
static int exec_line_vlite(orc_ctx *c, char *line, int lineno, int *last_id) {
    char *cut = strstr(line, ";;"); if (cut) *cut = 0;
    size_t L = strlen(line);
    while (L && isspace((unsigned char)line[L - 1])) line[--L] = 0;
    char *s = line; while (*s && isspace((unsigned char)*s)) s++;
    if (!*s) return 0;
    char *f[MAXF + 1]; memset(f, 0, sizeof f);
    int nf = split_fields(s, f, MAXF, 5);            /* a 5th field (Like's pattern) keeps its commas */
    if (nf < 3) return fail(c, "line %d: expected '<id>,<Op>,...'", lineno);
    const char *op = f[1];
    int64_t id64 = 0;
    const char *outname = "val";
    if (!strcmp(op, "Output") && parse_int(f[0], &id64)) { id64 = *last_id + 1; outname = f[0]; }
    else if (parse_int(f[0], &id64) || id64 <= 0 || id64 > (1 << 24)) return fail(c, "line %d: bad id", lineno);
    int id = (int)id64;
    *last_id = id;
    if (id >= c->nvecs) {
        int newn = id + 64;
        c->vecs = (ovec *)realloc(c->vecs, sizeof(ovec) * (size_t)newn);
        memset(c->vecs + c->nvecs, 0, sizeof(ovec) * (size_t)(newn - c->nvecs));
        c->nvecs = newn;
    }
    if (c->vecs[id].defined) return fail(c, "line %d: Id %d defined twice", lineno, id);
    ovec out; memset(&out, 0, sizeof out);
    int rc = -1, a, b, d;
    int64_t x, y, cnt;
    const ovec *va, *vb, *vd;
    c->ops_executed++;
#define NEED(k) if (nf != (k)) return fail(c, "line %d: %s expects %d fields, got %d", lineno, op, (k), nf)
#define REF1(k) if (parse_ref(f[k], &a) || !(va = get_vec(c, a))) return c->err[0] ? -1 : fail(c, "line %d: bad operand", lineno)
    if (!strcmp(op, "Load")) { NEED(3); rc = op_load(c, &out, f[2]); }
    else if (!strcmp(op, "Project") || !strcmp(op, "Shuffle")) { NEED(3); REF1(2); out = *va; out.owns = 0; rc = 0; }
    else if (!strcmp(op, "Semisort")) { NEED(3); REF1(2); rc = op_semisort(c, &out, va); }
    else if (!strcmp(op, "RangeV")) {
        NEED(5); if (parse_int(f[2], &x) || parse_int(f[4], &y)) return fail(c, "line %d: bad RangeV", lineno);
        REF1(3); rc = op_rangev(c, &out, x, va, y);
    } else if (!strcmp(op, "RangeC")) {
        NEED(5); if (parse_int(f[2], &x) || parse_int(f[3], &cnt) || parse_int(f[4], &y)) return fail(c, "line %d: bad RangeC", lineno);
        rc = op_rangec(c, &out, x, cnt, y);
    } else if (!strcmp(op, "Scatter")) {
        NEED(5); if (parse_ref(f[2], &a) || parse_ref(f[3], &b) || parse_ref(f[4], &d)) return fail(c, "line %d: bad Scatter", lineno);
        if (!(va = get_vec(c, a)) || !(vb = get_vec(c, b)) || !(vd = get_vec(c, d))) return -1;
        rc = op_scatter(c, &out, va, vb, vd);
    } else if (!strcmp(op, "Like")) {
        NEED(5); if (parse_ref(f[2], &a) || parse_ref(f[3], &b)) return fail(c, "line %d: bad operands for Like", lineno);
        if (!(va = get_vec(c, a)) || !(vb = get_vec(c, b))) return -1;
        rc = op_like(c, &out, va, vb, f[4]);
    } else if (!strcmp(op, "Output")) {
        if (nf != 3 && nf != 4) return fail(c, "line %d: Output expects 3 or 4 fields, got %d", lineno, nf);
        REF1(nf - 1);
        ovec named = *va; named.owns = 0;
        snprintf(named.field, sizeof named.field, "%s", outname);
        if (op_materialize(c, id, &named)) return -1;
        out = named; rc = 0;
    } else {
        NEED(4); if (parse_ref(f[2], &a) || parse_ref(f[3], &b)) return fail(c, "line %d: bad operands for %s", lineno, op);
        if (!(va = get_vec(c, a)) || !(vb = get_vec(c, b))) return -1;
        int bop = -1, fold = -1;
        for (int k = 0; k < OP_NBIN; k++) if (!strcmp(op, BIN_NAMES[k])) bop = k;
        if      (!strcmp(op, "FoldSelect")) fold = F_SEL;
        else if (!strcmp(op, "FoldSum"))    fold = F_SUM;
        else if (!strcmp(op, "FoldMin"))    fold = F_MIN;
        else if (!strcmp(op, "FoldMax"))    fold = F_MAX;
        else if (!strcmp(op, "FoldChoose")) fold = F_CHOOSE;
        else if (!strcmp(op, "FoldCount"))  fold = F_COUNT;
        if (bop >= 0) rc = op_binary(c, &out, bop, va, vb);
        else if (fold >= 0) rc = op_fold(c, &out, fold, va, vb);
        else if (!strcmp(op, "Partition")) rc = op_partition(c, &out, va, vb);
        else if (!strcmp(op, "Gather")) rc = op_gather(c, &out, va, vb);
        else if (!strcmp(op, "CrossProductOuter") || !strcmp(op, "CrossProductInner")) {
            if (vb->n > 0 && va->n > ((int64_t)1 << 24) / vb->n) return fail(c, "line %d: cross product of %lld x %lld slots", lineno, (long long)va->n, (long long)vb->n);
            if (alloc_vec(c, &out, va->n * vb->n, 0)) return -1;
            for (int64_t i = 0; i < out.n; i++) out.val[i] = op[12] == 'I' ? i % vb->n : i / vb->n;
            rc = 0;
        } else return fail(c, "line %d: unsupported operator '%s'", lineno, op);
    }
#undef NEED
#undef REF1
    if (rc) return -1;
    snprintf(out.field, sizeof out.field, "%s", !strcmp(op, "Output") ? outname : "val");
    out.defined = 1;
    c->vecs[id] = out;
    return 0;
}

int orc_run(orc_ctx *c, const char *text, size_t len) {
    clear_run_state(c);
    c->err[0] = 0; c->ops_executed = 0;
    char *buf = (char *)malloc(len + 1);
    memcpy(buf, text, len); buf[len] = 0;
    double t0 = now_s();
    int rc = 0, lineno = 0, last_id = 0;
    const int vlite = strstr(buf, ",Output,") != NULL;
    char *save = NULL;
    for (char *ln = strtok_r(buf, "\n", &save); ln; ln = strtok_r(NULL, "\n", &save)) {
        lineno++;
        if ((rc = vlite ? exec_line_vlite(c, ln, lineno, &last_id) : exec_line(c, ln, lineno))) break;
    }
    c->last_seconds = now_s() - t0;
    free(buf);
    /* intermediates are not needed after the run; outputs are kept */
    if (!c->keep_vectors) {
        for (int i = 0; i < c->nvecs; i++) free_vec(&c->vecs[i]);
        free(c->vecs); c->vecs = NULL; c->nvecs = 0;
    }
    return rc;
}

/* Tracing for the parity tests: with keep != 0 the vector of every statement stays readable after orc_run (until the
 * next run / orc_close), so that a mismatching program can be compared with the engine statement by statement.
 * orc_vector: *vals == NULL for a virtual range (value_i = from + i*step), *ok == NULL when every slot holds a value. */
void orc_keep_vectors(orc_ctx *c, int keep) { c->keep_vectors = keep; }
int orc_vector(const orc_ctx *c, int id, int64_t *n, const int64_t **vals, const uint8_t **ok, int64_t *from, int64_t *step) {
    if (id <= 0 || id >= c->nvecs || !c->vecs[id].defined) return -1;
    const ovec *v = &c->vecs[id];
    *n = v->n; *vals = v->is_range ? NULL : v->val; *ok = v->ok; *from = v->from; *step = v->step;
    return 0;
}

int orc_n_outputs(const orc_ctx *c) { return c->nouts; }

int orc_output(const orc_ctx *c, int k, const char **name, const char **tmp, const int64_t **vals, int64_t *n) {
    if (k < 0 || k >= c->nouts) return -1;
    *name = c->outs[k].name; *tmp = c->outs[k].tmp; *vals = c->outs[k].vals; *n = c->outs[k].n;
    return 0;
}

/* ------------------------------------------------------------------ synthetic data */
/* Counter-based generator of BASELINE.md / SURVEY.md section 8(d):
 *   v = add + mul * (lo + splitmix64(seed ^ col_id*PHI ^ row) mod (hi-lo+1))
 * Value ranges come from /root/reference/tests/tpch10noorder/bounds.csv:59-79. */
#define PHI 0x9E3779B97F4A7C15ULL
static inline uint64_t splitmix64(uint64_t x) {
    x += PHI;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
static inline int64_t gen_value(uint64_t seed, uint64_t col_id, uint64_t row, int64_t lo, uint64_t span, int64_t mul, int64_t add) {
    uint64_t h = splitmix64(seed ^ (col_id * PHI) ^ row);
    return (int64_t)((uint64_t)add + ((uint64_t)lo + h % span) * (uint64_t)mul);
}

uint64_t orc_col_id(const char *name) {          /* FNV-1a 64 of the column key path */
    uint64_t h = 0xCBF29CE484222325ULL;
    for (const unsigned char *p = (const unsigned char *)name; *p; p++) { h ^= *p; h *= 0x100000001B3ULL; }
    return h;
}

int orc_gen_column(void *out, int elem_bytes, int64_t row0, int64_t n, uint64_t seed, uint64_t col_id,
                   int64_t lo, int64_t hi, int64_t mul, int64_t add) {
    if (hi < lo) return -1;
    uint64_t span = (uint64_t)hi - (uint64_t)lo + 1;
    for (int64_t i = 0; i < n; i++) {
        int64_t v = gen_value(seed, col_id, (uint64_t)(row0 + i), lo, span, mul, add);
        switch (elem_bytes) {
        case 1: ((int8_t  *)out)[i] = (int8_t)v; break;
        case 2: ((int16_t *)out)[i] = (int16_t)v; break;
        case 4: ((int32_t *)out)[i] = (int32_t)v; break;
        case 8: ((int64_t *)out)[i] = v; break;
        default: return -1;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ SQL-semantics evaluators */
/* Independent of the interpreter above: written from the SQL in the plan headers.
 * Q6 (/root/reference/tests/tpch10noorder/06.sql.mplan:1-9):
 *   sum(l_extendedprice*l_discount) where shipdate in [1994-01-01, 1995-01-01),
 *   discount between 0.05 and 0.07, quantity < 24.
 * Encodings: dates = day counts (1994-01-01 = 728294, 1995-01-01 = 728659,
 * /root/reference/README.md:44,48), decimals(15,2) = value*100. */
int orc_sql_q6(const int32_t *shipdate, const int64_t *discount, const int64_t *quantity,
               const int64_t *extprice, int64_t n, int threads, int64_t *revenue, int64_t *count) {
    uint64_t rev = 0; int64_t cnt = 0;
#ifdef _OPENMP
    if (threads > 1) omp_set_num_threads(threads);
    #pragma omp parallel for reduction(+:rev,cnt) schedule(static) if (threads > 1)
#endif
    for (int64_t i = 0; i < n; i++) {
        if (shipdate[i] >= 728294 && shipdate[i] < 728659 && discount[i] >= 5 && discount[i] <= 7 && quantity[i] < 2400) {
            rev += (uint64_t)extprice[i] * (uint64_t)discount[i];
            cnt++;
        }
    }
    *revenue = (int64_t)rev; *count = cnt;
    (void)threads;
    return 0;
}

typedef struct { uint64_t seed; uint64_t col_id; int64_t lo, hi, mul, add; } orc_colspec;

/* Q6 over generated rows [row0,row0+n) without storing any column: lets bench.py
 * verify a full SF100 GPU answer bit-for-bit on the host cores in seconds.
 * specs order: shipdate, discount, quantity, extendedprice. */
int orc_sql_q6_generated(const orc_colspec *specs, int64_t row0, int64_t n, int threads,
                         int64_t *revenue, int64_t *count) {
    uint64_t rev = 0; int64_t cnt = 0;
    uint64_t span[4];
    for (int k = 0; k < 4; k++) span[k] = (uint64_t)specs[k].hi - (uint64_t)specs[k].lo + 1;
#ifdef _OPENMP
    if (threads > 1) omp_set_num_threads(threads);
    #pragma omp parallel for reduction(+:rev,cnt) schedule(static) if (threads > 1)
#endif
    for (int64_t i = 0; i < n; i++) {
        uint64_t row = (uint64_t)(row0 + i);
        int64_t sd = (int32_t)gen_value(specs[0].seed, specs[0].col_id, row, specs[0].lo, span[0], specs[0].mul, specs[0].add);
        if (sd < 728294 || sd >= 728659) continue;
        int64_t di = gen_value(specs[1].seed, specs[1].col_id, row, specs[1].lo, span[1], specs[1].mul, specs[1].add);
        if (di < 5 || di > 7) continue;
        int64_t qt = gen_value(specs[2].seed, specs[2].col_id, row, specs[2].lo, span[2], specs[2].mul, specs[2].add);
        if (qt >= 2400) continue;
        int64_t ep = gen_value(specs[3].seed, specs[3].col_id, row, specs[3].lo, span[3], specs[3].mul, specs[3].add);
        rev += (uint64_t)ep * (uint64_t)di; cnt++;
    }
    *revenue = (int64_t)rev; *count = cnt;
    (void)threads;
    return 0;
}

/* Q1 (/root/reference/tests/tpch10noorder/01.sql.mplan:1-18): group by (returnflag,
 * linestatus) where shipdate <= 1998-09-02 (= 729999, ordinal+365 per Mplan.hs:51-57).
 * Groups are reported in ascending (returnflag, linestatus) code order -- the order the
 * VDL program produces (sorted composite key).  out[g][0..9] = rf, ls, sum_qty,
 * sum_base_price, sum_disc_price, sum_charge, avg_qty, avg_price, avg_disc, count. */
int orc_sql_q1(const int32_t *shipdate, const int32_t *returnflag, const int32_t *linestatus,
               const int64_t *quantity, const int64_t *extprice, const int64_t *discount,
               const int64_t *tax, int64_t n, int64_t *out /* [max_groups][10] */, int max_groups, int *ngroups) {
    enum { MAXG = 4096 };
    static int64_t keys[MAXG][2]; static uint64_t acc[MAXG][6]; int ng = 0;
    for (int64_t i = 0; i < n; i++) {
        if (shipdate[i] > 729999) continue;
        int g = -1;
        for (int k = 0; k < ng; k++) if (keys[k][0] == returnflag[i] && keys[k][1] == linestatus[i]) { g = k; break; }
        if (g < 0) { if (ng == MAXG) return -1; g = ng++; keys[g][0] = returnflag[i]; keys[g][1] = linestatus[i]; memset(acc[g], 0, sizeof acc[g]); }
        uint64_t ep = (uint64_t)extprice[i], dp = ep * (uint64_t)(100 - discount[i]);
        acc[g][0] += (uint64_t)quantity[i];
        acc[g][1] += ep;
        acc[g][2] += dp;
        acc[g][3] += dp * (uint64_t)(100 + tax[i]);
        acc[g][4] += (uint64_t)discount[i];
        acc[g][5] += 1;
    }
    /* ascending key order */
    int order[MAXG];
    for (int k = 0; k < ng; k++) order[k] = k;
    for (int a = 1; a < ng; a++) { int t = order[a], b = a;
        while (b > 0 && (keys[order[b-1]][0] > keys[t][0] || (keys[order[b-1]][0] == keys[t][0] && keys[order[b-1]][1] > keys[t][1]))) { order[b] = order[b-1]; b--; }
        order[b] = t; }
    if (ng > max_groups) return -1;
    for (int r = 0; r < ng; r++) {
        int g = order[r]; int64_t *o = out + (int64_t)r * 10;
        int64_t cnt = (int64_t)acc[g][5];
        o[0] = keys[g][0]; o[1] = keys[g][1];
        o[2] = (int64_t)acc[g][0]; o[3] = (int64_t)acc[g][1]; o[4] = (int64_t)acc[g][2]; o[5] = (int64_t)acc[g][3];
        o[6] = w_div((int64_t)acc[g][0], cnt); o[7] = w_div((int64_t)acc[g][1], cnt); o[8] = w_div((int64_t)acc[g][4], cnt);
        o[9] = cnt;
    }
    *ngroups = ng;
    return 0;
}


/* Q1 over generated rows (no column storage), OpenMP over row chunks; same output layout as
 * orc_sql_q1.  specs order: shipdate, returnflag, linestatus, quantity, extendedprice, discount, tax. */
int orc_sql_q1_generated(const orc_colspec *specs, int64_t row0, int64_t n, int threads,
                         int64_t *out /* [max_groups][10] */, int max_groups, int *ngroups) {
    enum { NB = 32 };                       /* returnflag code / 8 in 2..8, linestatus code / 8 in 2..5 */
    uint64_t span[7];
    for (int k = 0; k < 7; k++) span[k] = (uint64_t)specs[k].hi - (uint64_t)specs[k].lo + 1;
    uint64_t acc[16][16][6];
    memset(acc, 0, sizeof acc);
#ifdef _OPENMP
    if (threads > 1) omp_set_num_threads(threads);
    #pragma omp parallel if (threads > 1)
#endif
    {
        uint64_t loc[16][16][6];
        memset(loc, 0, sizeof loc);
#ifdef _OPENMP
        #pragma omp for schedule(static)
#endif
        for (int64_t i = 0; i < n; i++) {
            uint64_t row = (uint64_t)(row0 + i);
            int64_t sd = gen_value(specs[0].seed, specs[0].col_id, row, specs[0].lo, span[0], specs[0].mul, specs[0].add);
            if (sd > 729999) continue;
            int64_t rf = gen_value(specs[1].seed, specs[1].col_id, row, specs[1].lo, span[1], specs[1].mul, specs[1].add);
            int64_t ls = gen_value(specs[2].seed, specs[2].col_id, row, specs[2].lo, span[2], specs[2].mul, specs[2].add);
            int64_t qt = gen_value(specs[3].seed, specs[3].col_id, row, specs[3].lo, span[3], specs[3].mul, specs[3].add);
            int64_t ep = gen_value(specs[4].seed, specs[4].col_id, row, specs[4].lo, span[4], specs[4].mul, specs[4].add);
            int64_t di = gen_value(specs[5].seed, specs[5].col_id, row, specs[5].lo, span[5], specs[5].mul, specs[5].add);
            int64_t tx = gen_value(specs[6].seed, specs[6].col_id, row, specs[6].lo, span[6], specs[6].mul, specs[6].add);
            uint64_t *a = loc[(rf >> 3) & 15][(ls >> 3) & 15];
            uint64_t dp = (uint64_t)ep * (uint64_t)(100 - di);
            a[0] += (uint64_t)qt; a[1] += (uint64_t)ep; a[2] += dp; a[3] += dp * (uint64_t)(100 + tx); a[4] += (uint64_t)di; a[5] += 1;
        }
#ifdef _OPENMP
        #pragma omp critical
#endif
        for (int x = 0; x < 16; x++) for (int y = 0; y < 16; y++) for (int k = 0; k < 6; k++) acc[x][y][k] += loc[x][y][k];
    }
    int ng = 0;
    for (int x = 0; x < 16; x++) for (int y = 0; y < 16; y++) {
        if (!acc[x][y][5]) continue;
        if (ng == max_groups) return -1;
        int64_t *o = out + (int64_t)ng * 10; int64_t cnt = (int64_t)acc[x][y][5];
        o[0] = x * 8; o[1] = y * 8;
        o[2] = (int64_t)acc[x][y][0]; o[3] = (int64_t)acc[x][y][1]; o[4] = (int64_t)acc[x][y][2]; o[5] = (int64_t)acc[x][y][3];
        o[6] = w_div((int64_t)acc[x][y][0], cnt); o[7] = w_div((int64_t)acc[x][y][1], cnt); o[8] = w_div((int64_t)acc[x][y][4], cnt);
        o[9] = cnt; ng++;
    }
    *ngroups = ng;
    (void)threads; (void)NB;
    return 0;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* q6_device.c -- a plain C host driving libvdl end to end on the GPU, the way a Haskell host would through
 * `foreign import ccall`: generate the four Q6 columns in HBM, parse the VDL text the
 * compiler prints for tests/tpch10noorder/06.sql.mplan, run it, read the answer back with vdl_output.
 *
 *   gcc -std=c11 -Iinclude examples/q6_device.c -Lmplan2vdl_amd/lib -lvdl -Wl,-rpath,$PWD/mplan2vdl_amd/lib -o q6_device
 *   ./q6_device tests/golden/q6.vdl 60175 [0|1 = fuse]
 *
 * Prints the reply document of /root/reference/resolve.py:8-32 on stdout.  tests/test_pipe_end.py compiles and runs
 * it on the device and compares the answer with the oracle's. */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>

#include "vdl.h"

/* value ranges: /root/reference/tests/tpch10noorder/bounds.csv:59-79; v = add + mul * (lo + hash(row) % (hi - lo + 1)) */
static const struct { const char *name; int width; int64_t lo, hi, mul, add; } kCols[] = {
    {"lineitem.l_shipdate", 4, 727564, 730089, 1, 0},
    {"lineitem.l_discount", 8, 0, 10, 1, 0},
    {"lineitem.l_quantity", 8, 1, 50, 100, 0},
    {"lineitem.l_extendedprice", 8, 90091, 10494950, 1, 0},
};

static int fail(vdl_ctx *c, const char *what, int rc) {
    fprintf(stderr, "q6_device: %s failed (%d): %s\n", what, rc, c ? vdl_last_error(c) : "");
    return 1;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: q6_device program.vdl rows [fuse]\n"); return 2; }
    const int64_t rows = atoll(argv[2]);
    const int fuse = argc > 3 ? atoi(argv[3]) : 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *text = (char *)malloc((size_t)len + 1);
    if (!text || fread(text, 1, (size_t)len, f) != (size_t)len) { fprintf(stderr, "q6_device: cannot read %s\n", argv[1]); return 2; }
    fclose(f);

    vdl_ctx *ctx = NULL;
    int rc = vdl_open(&ctx, 0);
    if (rc) return fail(ctx, "vdl_open", rc);
    for (size_t k = 0; k < sizeof kCols / sizeof kCols[0]; k++)
        if ((rc = vdl_generate_column(ctx, kCols[k].name, kCols[k].width, 0, rows, 0x5EED0006ULL, kCols[k].lo, kCols[k].hi, kCols[k].mul, kCols[k].add)))
            return fail(ctx, "vdl_generate_column", rc);
    vdl_plan *plan = NULL;
    if ((rc = vdl_parse(ctx, text, (size_t)len, &plan))) return fail(ctx, "vdl_parse", rc);
    vdl_plan_set_fusion(plan, fuse);
    if (vdl_plan_is_fused(plan) != (fuse != 0)) { fprintf(stderr, "q6_device: plan fused = %d, asked for %d\n", vdl_plan_is_fused(plan), fuse); return 1; }
    if ((rc = vdl_run(ctx, plan))) return fail(ctx, "vdl_run", rc);
    printf("{\"results\": {");
    for (int k = 0; k < vdl_n_outputs(plan); k++) {
        const char *name, *tmp;
        const int64_t *vals;
        size_t n;
        if ((rc = vdl_output(plan, k, &name, &tmp, &vals, &n))) return fail(ctx, "vdl_output", rc);
        printf("%s\"%s\": {\".%s\": [", k ? ", " : "", tmp, name);
        for (size_t i = 0; i < n; i++) printf("%s%" PRId64, i ? ", " : "", vals[i]);
        printf("]}");
    }
    printf("}, \"timings\": {}}\n");
    vdl_plan_free(plan);
    vdl_close(ctx);
    free(text);
    return 0;
}

#include <stdio.h>
#include "vdl.h"
int main(void) {
    vdl_ctx *ctx = NULL;
    if (vdl_open(&ctx, -1)) return 1;                 /* host-only context: parse / describe */
    const char text[] = "1,Load,t.a\n2,Project,val,Id 1,a\n3,MaterializeCompact,Id 2\n";
    vdl_plan *plan = NULL;
    if (vdl_parse(ctx, text, sizeof text - 1, &plan)) { fprintf(stderr, "%s\n", vdl_last_error(ctx)); return 1; }
    fputs(vdl_plan_describe(plan), stdout);
    int rc = vdl_run(ctx, plan);                      /* no device: must fail loudly */
    printf("vdl_run without a device -> %d (%s)\n", rc, vdl_last_error(ctx));
    vdl_plan_free(plan);
    vdl_close(ctx);
    return rc == VDL_ERR_DEVICE ? 0 : 1;
}

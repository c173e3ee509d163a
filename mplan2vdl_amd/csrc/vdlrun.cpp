// vdlrun -- the executor end of the reference's pipe:
//   ./tpchrun DIR plan.mplan | sed 's/;;.*//' | vdlrun --rows N            (synthetic TPC-H-shaped lineitem)
// reads VDL text on stdin, runs it on the GPU through libvdl and prints the JSON document that
// /root/reference/resolve.py:8-32 consumes: {"results": {"tmpN": {".name": [...]}}, "timings": {...}}.
// Exit status is non-zero on any error (resolve.py:42-43 treats a missing "results" key as failure).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "vdl.h"

namespace {
struct GenSpec { const char *name; int width; int64_t lo, hi, mul, add; };
// value ranges: /root/reference/tests/tpch10noorder/bounds.csv:59-79
const GenSpec kLineitem[] = {
    {"lineitem.l_shipdate", 4, 727564, 730089, 1, 0},      {"lineitem.l_discount", 8, 0, 10, 1, 0},
    {"lineitem.l_quantity", 8, 1, 50, 100, 0},             {"lineitem.l_extendedprice", 8, 90091, 10494950, 1, 0},
    {"lineitem.l_tax", 8, 0, 8, 1, 0},                     {"lineitem.l_returnflag", 4, 0, 2, 24, 16},
    {"lineitem.l_linestatus", 4, 0, 1, 24, 16},
};

int die(vdl_ctx *c, const char *what, int rc) {
    std::fprintf(stderr, "vdlrun: %s failed (%d): %s\n", what, rc, c ? vdl_last_error(c) : "");
    return 1;
}
}  // namespace

int main(int argc, char **argv) {
    int64_t rows = 60175;           // SF0.01 lineitem, /root/reference/tests/tpchnoorder/bounds.csv:59
    uint64_t seed = 0x5EED0006ULL;
    int device = 0, fuse = 1, profile = 0, describe = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--rows" && i + 1 < argc) rows = std::atoll(argv[++i]);
        else if (a == "--seed" && i + 1 < argc) seed = std::strtoull(argv[++i], nullptr, 0);
        else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--no-fuse") fuse = 0;
        else if (a == "--profile") profile = 1;
        else if (a == "--describe") describe = 1;
        else { std::fprintf(stderr, "usage: vdlrun [--rows N] [--seed S] [--device D] [--no-fuse] [--profile] [--describe] < program.vdl\n"); return 2; }
    }
    std::string text((std::istreambuf_iterator<char>(std::cin)), std::istreambuf_iterator<char>());
    vdl_ctx *ctx = nullptr;
    int rc = vdl_open(&ctx, describe ? -1 : device);
    if (rc) return die(ctx, "vdl_open", rc);
    vdl_plan *plan = nullptr;
    if ((rc = vdl_parse(ctx, text.data(), text.size(), &plan))) return die(ctx, "vdl_parse", rc);
    vdl_plan_set_fusion(plan, fuse);
    vdl_plan_set_profiling(plan, profile);
    if (describe) { std::fputs(vdl_plan_describe(plan), stdout); return 0; }
    for (const GenSpec &g : kLineitem)
        if ((rc = vdl_generate_column(ctx, g.name, g.width, 0, rows, seed, g.lo, g.hi, g.mul, g.add))) return die(ctx, "vdl_generate_column", rc);
    if ((rc = vdl_run(ctx, plan))) return die(ctx, "vdl_run", rc);
    std::printf("{\"results\": {");
    for (int k = 0; k < vdl_n_outputs(plan); k++) {
        const char *name, *tmp; const int64_t *vals; size_t n;
        vdl_output(plan, k, &name, &tmp, &vals, &n);
        std::printf("%s\"%s\": {\".%s\": [", k ? ", " : "", tmp, name);
        for (size_t i = 0; i < n; i++) std::printf("%s%lld", i ? ", " : "", (long long)vals[i]);
        std::printf("]}");
    }
    std::printf("}, \"timings\": {");
    for (int k = 0; k < vdl_n_timings(plan); k++) {
        const char *label; double us;
        vdl_timing(plan, k, &label, &us);
        std::printf("%s\"%s\": %.0f", k ? ", " : "", label, us);
    }
    std::printf("}}\n");
    vdl_plan_free(plan);
    vdl_close(ctx);
    return 0;
}

// vdlrun -- the executor end of the reference's pipe:
//   ./tpchrun DIR plan.mplan | sed 's/;;.*//' | vdlrun --rows N            (synthetic TPC-H-shaped lineitem)
//   ./tpchrun DIR plan.mplan | sed 's/;;.*//' | vdlrun --data COLDIR       (exported columns, see below)
// reads VDL text on stdin, runs it on the GPU through libvdl and prints the JSON document that
// /root/reference/resolve.py:8-32 consumes: {"results": {"tmpN": {".name": [...]}}, "timings": {...}}.
// Exit status is non-zero on any error (resolve.py:42-43 treats a missing "results" key as failure).
// --data COLDIR: COLDIR/columns.csv lists "<table.col>,<bytes per element>,<rows>" and COLDIR/<table.col>.bin holds
// the raw little-endian array (what mplan2vdl_amd.catalog.export_columns writes; a MonetDB BAT tail file of a
// fixed-width column has the same layout).  Only the columns the program Loads are read.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
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

struct ColFile { int width; int64_t rows; };

int load_data_dir(vdl_ctx *ctx, const std::string &dir, const std::string &program) {
    std::map<std::string, ColFile> listed;
    std::ifstream manifest(dir + "/columns.csv");
    if (!manifest) { std::fprintf(stderr, "vdlrun: cannot read %s/columns.csv\n", dir.c_str()); return 1; }
    std::string line;
    while (std::getline(manifest, line)) {
        std::istringstream ls(line);
        std::string name, w, r;
        if (!std::getline(ls, name, ',') || !std::getline(ls, w, ',') || !std::getline(ls, r, ',')) continue;
        listed[name] = ColFile{std::atoi(w.c_str()), std::atoll(r.c_str())};
    }
    std::istringstream prog(program);
    while (std::getline(prog, line)) {
        const size_t at = line.find(",Load,");
        if (at == std::string::npos) continue;
        std::string name = line.substr(at + 6);
        const size_t cut = name.find(";;");
        if (cut != std::string::npos) name.resize(cut);
        while (!name.empty() && isspace((unsigned char)name.back())) name.pop_back();
        auto it = listed.find(name);
        if (it == listed.end()) { std::fprintf(stderr, "vdlrun: column %s is not in %s/columns.csv\n", name.c_str(), dir.c_str()); return 1; }
        const size_t bytes = (size_t)it->second.width * (size_t)it->second.rows;
        std::vector<char> buf(bytes ? bytes : 1);
        std::ifstream f(dir + "/" + name + ".bin", std::ios::binary);
        if (!f || (bytes && !f.read(buf.data(), (std::streamsize)bytes))) {
            std::fprintf(stderr, "vdlrun: %s/%s.bin is missing or shorter than %zu bytes\n", dir.c_str(), name.c_str(), bytes);
            return 1;
        }
        const int rc = vdl_upload_column(ctx, name.c_str(), buf.data(), it->second.width, it->second.rows);
        if (rc) { std::fprintf(stderr, "vdlrun: vdl_upload_column(%s) failed (%d): %s\n", name.c_str(), rc, vdl_last_error(ctx)); return 1; }
        listed.erase(it);                                  // a column loaded twice by the program is uploaded once
        listed[name] = ColFile{0, -1};
    }
    return 0;
}

int die(vdl_ctx *c, const char *what, int rc) {
    std::fprintf(stderr, "vdlrun: %s failed (%d): %s\n", what, rc, c ? vdl_last_error(c) : "");
    return 1;
}
}  // namespace

int main(int argc, char **argv) {
    int64_t rows = 60175;           // SF0.01 lineitem, /root/reference/tests/tpchnoorder/bounds.csv:59
    uint64_t seed = 0x5EED0006ULL;
    int device = 0, fuse = 1, profile = 0, describe = 0;
    std::string data_dir;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--rows" && i + 1 < argc) rows = std::atoll(argv[++i]);
        else if (a == "--data" && i + 1 < argc) data_dir = argv[++i];
        else if (a == "--seed" && i + 1 < argc) seed = std::strtoull(argv[++i], nullptr, 0);
        else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--no-fuse") fuse = 0;
        else if (a == "--profile") profile = 1;
        else if (a == "--describe") describe = 1;
        else { std::fprintf(stderr, "usage: vdlrun [--rows N | --data DIR] [--seed S] [--device D] [--no-fuse] [--profile] [--describe] < program.vdl\n"); return 2; }
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
    if (!data_dir.empty()) {
        if (load_data_dir(ctx, data_dir, text)) return 1;
    } else {
        for (const GenSpec &g : kLineitem)
            if ((rc = vdl_generate_column(ctx, g.name, g.width, 0, rows, seed, g.lo, g.hi, g.mul, g.add))) return die(ctx, "vdl_generate_column", rc);
    }
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

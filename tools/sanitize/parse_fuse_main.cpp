// CPU-only sanitizer harness: parses and plans every VDL file given on the command line plus a set of
// mutated / malformed variants, under AddressSanitizer + UBSan (tools/sanitize/run.sh).
// GPU sanitizers are not available on this pool, so the host-side front door is checked this way.
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>

#include <set>

#include "vdl.h"
#include "vdl_exchange_analysis.h"
#include "vdl_fuse.h"
#include "vdl_ir.h"

static long routes_found = 0;
// the sharded routes' analyses (which vectors travel, why a program does or does not qualify) for every table the program loads
static void analyse_routes(const vdl::Program &p) {
    std::set<std::string> tables;
    for (int id : p.order) {
        const vdl::Node &n = p.at(id);
        if (n.op == vdl::Op::Load) { const size_t dot = n.column.find('.'); if (dot != std::string::npos) tables.insert(n.column.substr(0, dot)); }
    }
    tables.insert("");
    for (const std::string &t : tables) {
        routes_found += vdl::exan::analyse_exchange(p, t, true).ok;
        routes_found += vdl::exan::analyse_exchange(p, t, false, true).ok;
        routes_found += vdl::exan::analyse_folds(p, t).ok;
        routes_found += vdl::exan::analyse_chain(p, t).ok;
    }
}

static int try_one(const std::string &text) {
    try {
        vdl::Program p = vdl::parse_program(text.data(), text.size());
        vdl::rewrite_program(p);
        analyse_routes(p);
        vdl::FusedPlan f = vdl::fuse_program(p);
        return (int)vdl::describe_fused(f).size() > 0 ? 0 : 1;
    } catch (const vdl::Error &e) {
        return e.code;
    }
}

int main(int argc, char **argv) {
    int ok = 0, rejected = 0;
    for (int i = 1; i < argc; i++) {
        std::ifstream in(argv[i]);
        std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (try_one(text) == 0) ok++; else rejected++;
        // truncations and single-character corruptions of the program
        for (size_t cut = 0; cut < text.size(); cut += 7) (try_one(text.substr(0, cut)) == 0 ? ok : rejected)++;
        for (size_t k = 0; k < text.size(); k += 11) {
            std::string m = text;
            m[k] = (k % 3 == 0) ? ',' : (k % 3 == 1) ? '9' : 'I';
            (try_one(m) == 0 ? ok : rejected)++;
        }
    }
    std::printf("sanitize: %d programs planned, %d rejected with an error code, %ld sharded routes found, no sanitizer report\n", ok, rejected, routes_found);
    return 0;
}

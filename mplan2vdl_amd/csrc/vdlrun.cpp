// vdlrun -- the executor end of the reference's pipe:
//   ./tpchrun DIR plan.mplan | sed 's/;;.*//' | vdlrun --rows N            (synthetic TPC-H-shaped lineitem)
//   ./tpchrun DIR plan.mplan | sed 's/;;.*//' | vdlrun --data COLDIR       (exported columns, see below)
// reads VDL text on stdin, runs it on the GPU through libvdl and prints the JSON document that
// /root/reference/resolve.py:8-32 consumes: {"results": {"tmpN": {".name": [...]}}, "timings": {...}}.
// Exit status is non-zero on any error (resolve.py:42-43 treats a missing "results" key as failure).
// --data COLDIR: COLDIR/columns.csv lists "<table.col>,<bytes per element>,<rows>" and COLDIR/<table.col>.bin holds
// the raw little-endian array (what mplan2vdl_amd.catalog.export_columns writes; a MonetDB BAT tail file of a
// fixed-width column has the same layout).  Only the columns the program Loads are read.
//   ... | vdlrun --gpus N [--shard TABLE] --rows M | --data COLDIR        one process per GPU (SURVEY.md section 8(b),(e))
// --gpus N forks N ranks BEFORE anything touches HIP; rank r opens device r, holds rows [r*n/N, (r+1)*n/N) of the sharded
// table (--shard, default lineitem) and the other tables in full, joins an RCCL communicator (rank 0's id travels through a
// file in a private temporary directory) and calls vdl_run_sharded.  The parent prints ONE reply: the common answer of a
// fold plan, or the ranks' slices concatenated in rank order for a plan with a Partition.
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

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

int load_data_dir(vdl_ctx *ctx, const std::string &dir, const std::string &program, const std::string &shard_table = "", int rank = 0, int world = 1,
                  int64_t *row0_out = nullptr) {
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
        // a column of the sharded table: this rank's row range only (string heaps "table.col.heap" are not row-aligned: whole)
        int64_t lo = 0, hi = it->second.rows;
        const bool is_heap = name.size() > 5 && name.compare(name.size() - 5, 5, ".heap") == 0;
        if (world > 1 && !shard_table.empty() && !is_heap && name.compare(0, shard_table.size() + 1, shard_table + ".") == 0) {
            lo = it->second.rows * rank / world; hi = it->second.rows * (rank + 1) / world;
            if (row0_out) *row0_out = lo;
        }
        const size_t bytes = (size_t)it->second.width * (size_t)(hi - lo);
        std::vector<char> buf(bytes ? bytes : 1);
        std::ifstream f(dir + "/" + name + ".bin", std::ios::binary);
        if (f) f.seekg((std::streamoff)((size_t)it->second.width * (size_t)lo));
        if (!f || (bytes && !f.read(buf.data(), (std::streamsize)bytes))) {
            std::fprintf(stderr, "vdlrun: %s/%s.bin is missing or shorter than the %lld rows columns.csv lists\n", dir.c_str(), name.c_str(), (long long)it->second.rows);
            return 1;
        }
        const int rc = vdl_upload_column(ctx, name.c_str(), buf.data(), it->second.width, hi - lo);
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
struct Reply {
    struct Out { std::string name, tmp; std::vector<int64_t> vals; };
    std::vector<Out> outs;
    std::vector<std::pair<std::string, double>> timings;
    bool replicated = true;          // every rank holds the whole answer (fold plans) / this is one rank's slice (Partition plans)
};

void collect(vdl_plan *plan, Reply &r) {
    for (int k = 0; k < vdl_n_outputs(plan); k++) {
        const char *name, *tmp; const int64_t *vals; size_t n;
        vdl_output(plan, k, &name, &tmp, &vals, &n);
        r.outs.push_back({name, tmp, std::vector<int64_t>(vals, vals + n)});
    }
    for (int k = 0; k < vdl_n_timings(plan); k++) {
        const char *label; double us;
        vdl_timing(plan, k, &label, &us);
        r.timings.push_back({label, us});
    }
}

void print_reply(const Reply &r) {
    std::printf("{\"results\": {");
    for (size_t k = 0; k < r.outs.size(); k++) {
        std::printf("%s\"%s\": {\".%s\": [", k ? ", " : "", r.outs[k].tmp.c_str(), r.outs[k].name.c_str());
        for (size_t i = 0; i < r.outs[k].vals.size(); i++) std::printf("%s%lld", i ? ", " : "", (long long)r.outs[k].vals[i]);
        std::printf("]}");
    }
    std::printf("}, \"timings\": {");
    for (size_t k = 0; k < r.timings.size(); k++) std::printf("%s\"%s\": %.0f", k ? ", " : "", r.timings[k].first.c_str(), r.timings[k].second);
    std::printf("}}\n");
}

bool write_reply(const std::string &path, const Reply &r) {
    std::ofstream f(path + ".tmp", std::ios::binary);
    f << (r.replicated ? 1 : 0) << "\n" << r.outs.size() << "\n";
    for (const auto &o : r.outs) {
        f << o.name << "\n" << o.tmp << "\n" << o.vals.size() << "\n";
        f.write((const char *)o.vals.data(), (std::streamsize)(sizeof(int64_t) * o.vals.size()));
        f << "\n";
    }
    f << r.timings.size() << "\n";
    for (const auto &t : r.timings) f << t.first << "\n" << t.second << "\n";
    f.close();
    return f.good() && std::rename((path + ".tmp").c_str(), path.c_str()) == 0;
}

bool read_reply(const std::string &path, Reply &r) {
    std::ifstream f(path, std::ios::binary);
    std::string line;
    auto num = [&]() { std::getline(f, line); return std::atoll(line.c_str()); };
    if (!f) return false;
    r.replicated = num() != 0;
    const long long nouts = num();
    for (long long k = 0; k < nouts; k++) {
        Reply::Out o;
        std::getline(f, o.name); std::getline(f, o.tmp);
        o.vals.resize((size_t)num());
        f.read((char *)o.vals.data(), (std::streamsize)(sizeof(int64_t) * o.vals.size()));
        std::getline(f, line);
        r.outs.push_back(std::move(o));
    }
    const long long nt = num();
    for (long long k = 0; k < nt; k++) { std::string label; std::getline(f, label); std::getline(f, line); r.timings.push_back({label, std::atof(line.c_str())}); }
    return (bool)f || f.eof();
}

struct Options {
    int64_t rows = 60175;           // SF0.01 lineitem, /root/reference/tests/tpchnoorder/bounds.csv:59
    uint64_t seed = 0x5EED0006ULL;
    int device = 0, fuse = 1, profile = 0, describe = 0, gpus = 1, jit = 0;
    std::string data_dir, shard = "lineitem";
};

// one rank of `world` (world = 1 without --gpus: plain vdl_run); comm_dir = where rank 0 leaves the communicator id
int run_rank(const Options &o, const std::string &text, int rank, int world, const std::string &comm_dir, Reply &reply) {
    vdl_ctx *ctx = nullptr;
    int rc = vdl_open(&ctx, o.describe ? -1 : (world > 1 || !comm_dir.empty() ? rank : o.device));
    if (rc) return die(ctx, "vdl_open", rc);
    vdl_plan *plan = nullptr;
    if ((rc = vdl_parse(ctx, text.data(), text.size(), &plan))) return die(ctx, "vdl_parse", rc);
    vdl_plan_set_fusion(plan, o.fuse);
    vdl_plan_set_profiling(plan, o.profile);
    if (o.jit) vdl_plan_set_jit(plan, o.jit);
    if (o.describe) { std::fputs(vdl_plan_describe(plan), stdout); return 0; }
    int64_t row0 = 0;
    if (!o.data_dir.empty()) {
        if (load_data_dir(ctx, o.data_dir, text, o.shard, rank, world, &row0)) return 1;
    } else {
        row0 = o.rows * rank / world;
        const int64_t mine = o.rows * (rank + 1) / world - row0;
        for (const GenSpec &g : kLineitem)
            if ((rc = vdl_generate_column(ctx, g.name, g.width, row0, mine, o.seed, g.lo, g.hi, g.mul, g.add))) return die(ctx, "vdl_generate_column", rc);
    }
    if (comm_dir.empty()) {
        if ((rc = vdl_run(ctx, plan))) return die(ctx, "vdl_run", rc);
    } else {
        unsigned char id[VDL_COMM_ID_BYTES];
        const std::string id_path = comm_dir + "/id";
        if (rank == 0) {
            if ((rc = vdl_comm_unique_id(id))) return die(ctx, "vdl_comm_unique_id", rc);
            std::ofstream f(id_path + ".tmp", std::ios::binary);
            f.write((const char *)id, sizeof id);
            f.close();
            if (!f.good() || std::rename((id_path + ".tmp").c_str(), id_path.c_str())) { std::fprintf(stderr, "vdlrun: cannot write %s\n", id_path.c_str()); return 1; }
        } else {
            bool got = false;
            for (int tries = 0; tries < 6000 && !got; tries++) {          // up to 60 s for rank 0 to get there
                std::ifstream f(id_path, std::ios::binary);
                got = f && f.read((char *)id, sizeof id);
                if (!got) usleep(10000);
            }
            if (!got) { std::fprintf(stderr, "vdlrun: rank %d never saw the communicator id\n", rank); return 1; }
        }
        if ((rc = vdl_comm_init(ctx, rank, world, id))) return die(ctx, "vdl_comm_init", rc);
        vdl_plan_set_sharded_table(plan, o.shard.c_str());
        vdl_plan_set_row_offset(plan, row0);
        int whole = 0;
        const char *route = nullptr;
        if ((rc = vdl_plan_sharded_route(ctx, plan, &route, &whole))) return die(ctx, "vdl_plan_sharded_route", rc);
        reply.replicated = whole != 0;
        if ((rc = vdl_run_sharded(ctx, plan))) return die(ctx, "vdl_run_sharded", rc);
    }
    collect(plan, reply);
    vdl_plan_free(plan);
    vdl_close(ctx);
    return 0;
}
}  // namespace

int main(int argc, char **argv) {
    Options o;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--rows" && i + 1 < argc) o.rows = std::atoll(argv[++i]);
        else if (a == "--data" && i + 1 < argc) o.data_dir = argv[++i];
        else if (a == "--seed" && i + 1 < argc) o.seed = std::strtoull(argv[++i], nullptr, 0);
        else if (a == "--device" && i + 1 < argc) o.device = std::atoi(argv[++i]);
        else if (a == "--gpus" && i + 1 < argc) o.gpus = std::atoi(argv[++i]);
        else if (a == "--shard" && i + 1 < argc) o.shard = argv[++i];
        else if (a == "--no-fuse") o.fuse = 0;
        else if (a == "--jit") o.jit = 1;
        else if (a == "--jit-tune") o.jit = 2;
        else if (a == "--profile") o.profile = 1;
        else if (a == "--describe") o.describe = 1;
        else {
            std::fprintf(stderr, "usage: vdlrun [--rows N | --data DIR] [--gpus N [--shard TABLE]] [--seed S] [--device D] [--no-fuse] [--jit | --jit-tune] [--profile] [--describe] < program.vdl\n");
            return 2;
        }
    }
    if (o.gpus < 1 || o.gpus > 128) { std::fprintf(stderr, "vdlrun: --gpus must be 1 .. 128\n"); return 2; }
    std::string text((std::istreambuf_iterator<char>(std::cin)), std::istreambuf_iterator<char>());
    bool sharded = false;
    for (int i = 1; i < argc; i++) sharded = sharded || std::string(argv[i]) == "--gpus";
    if (!sharded || o.describe) {
        Reply r;
        const int rc = run_rank(o, text, 0, 1, "", r);
        if (rc == 0 && !o.describe) print_reply(r);
        return rc;
    }
    // one process per GPU, forked before anything in this process has touched HIP (no HIP call above this line)
    char tmpl[] = "/tmp/vdlrun.XXXXXX";
    const char *dir = mkdtemp(tmpl);
    if (!dir) { std::perror("vdlrun: mkdtemp"); return 1; }
    std::vector<pid_t> kids;
    for (int r = 0; r < o.gpus; r++) {
        const pid_t pid = fork();
        if (pid < 0) { std::perror("vdlrun: fork"); return 1; }
        if (pid == 0) {
            dup2(2, 1);                                      // stdout carries the parent's reply only (RCCL prints a banner at start-up)
            Reply mine;
            int rc = run_rank(o, text, r, o.gpus, dir, mine);
            if (rc == 0 && !write_reply(std::string(dir) + "/out." + std::to_string(r), mine)) rc = 1;
            std::fflush(nullptr);
            _exit(rc);
        }
        kids.push_back(pid);
    }
    // a rank that fails leaves its peers waiting in the communicator: the first failure ends the others
    int failed = 0;
    for (size_t left = kids.size(); left > 0; left--) {
        int st = 0;
        const pid_t pid = waitpid(-1, &st, 0);
        if (pid < 0) { failed++; break; }
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
            if (!failed)
                for (pid_t other : kids) if (other != pid) kill(other, SIGTERM);
            failed++;
        }
    }
    Reply all;
    for (int r = 0; r < o.gpus && !failed; r++) {
        Reply part;
        if (!read_reply(std::string(dir) + "/out." + std::to_string(r), part)) { failed++; break; }
        if (r == 0) { all = part; continue; }
        if (part.replicated) continue;                       // fold plans: every rank printed the same answer
        for (size_t k = 0; k < all.outs.size() && k < part.outs.size(); k++)
            all.outs[k].vals.insert(all.outs[k].vals.end(), part.outs[k].vals.begin(), part.outs[k].vals.end());
    }
    for (int r = 0; r < o.gpus; r++) std::remove((std::string(dir) + "/out." + std::to_string(r)).c_str());
    std::remove((std::string(dir) + "/id").c_str());
    rmdir(dir);
    if (failed) { std::fprintf(stderr, "vdlrun: %d rank(s) failed\n", failed); return 1; }
    print_reply(all);
    return 0;
}

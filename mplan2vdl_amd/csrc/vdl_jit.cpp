// vdl_jit.cpp -- a fused scan specialised for ONE plan at run time.
//
// The precompiled scan kernels (vdl_mscan.hip) interpret a descriptor: which columns are lookups, which carry filters, the
// group key's steps, every aggregate's factors, the conditions' tests.  That costs vector instructions per row -- selects
// over all columns to find a source column, dispatch on kinds -- and a 28 B/row scan at 8 TB/s has only about 140 of them
// per row slice of a wave.  Here the SAME device code (vdl_mscan_body.h, embedded as text at build time) is compiled by
// hiprtc with the plan's descriptor as a compile-time constant: every descriptor-driven branch and loop folds away and
// what is left is the straight-line code of this query.  (The text this engine executes "is meant to be executed by a voodoo
// implementation", /root/reference/README.md:57 -- the system of the cited paper, which generates kernels per program.)
// One hiprtc compile per distinct (descriptor, shape) per process -- seconds -- cached in memory and, when VDL_JIT_CACHE names a directory, on disk.  Opt-in: vdl_plan_set_jit / VDL_JIT=1; a plan
// whose specialisation fails to build runs on the precompiled kernels and says so in its description.
#include "vdl_jit.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>

namespace vdl {
namespace jit {

namespace {

#ifndef VDL_PROJ_U
#define VDL_PROJ_U 4
#endif
constexpr int VDL_PROJ_U_HOST = VDL_PROJ_U;
#ifndef VDL_FRONT_BATCH
#define VDL_FRONT_BATCH 4
#endif
constexpr int VDL_FRONT_BATCH_HOST = VDL_FRONT_BATCH;      // (tiles per batch of the one-pass front as this library was built: the launcher's grid and look-back area follow it)
const char kEmbedded[] =
#include "vdl_jit_src.inc"
    ;

// ---- hiprtc, bound at first use ------------------------------------------------------------------------------------------
struct Rtc {
    void *lib = nullptr;
    int (*create)(void **, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*compile)(void *, int, const char **) = nullptr;
    int (*log_size)(void *, size_t *) = nullptr;
    int (*log)(void *, char *) = nullptr;
    int (*code_size)(void *, size_t *) = nullptr;
    int (*code)(void *, char *) = nullptr;
    int (*destroy)(void **) = nullptr;
    int (*version)(int *, int *) = nullptr;      // optional
    std::string why;
};
Rtc &rtc() {
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
        if (const char *forced = getenv("VDL_HIPRTC_LIB")) r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);      // (this one or none)
        else for (const char *n : names) if ((r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!r.lib) { r.why = "libhiprtc not found"; return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p) r.why = std::string("libhiprtc lacks ") + n; return p; };
        r.create = (decltype(r.create))sym("hiprtcCreateProgram");
        r.compile = (decltype(r.compile))sym("hiprtcCompileProgram");
        r.log_size = (decltype(r.log_size))sym("hiprtcGetProgramLogSize");
        r.log = (decltype(r.log))sym("hiprtcGetProgramLog");
        r.code_size = (decltype(r.code_size))sym("hiprtcGetCodeSize");
        r.code = (decltype(r.code))sym("hiprtcGetCode");
        r.destroy = (decltype(r.destroy))sym("hiprtcDestroyProgram");
        r.version = (decltype(r.version))dlsym(r.lib, "hiprtcVersion");
    });
    return r;
}

std::string lit(int64_t v) {
    if (v == INT64_MIN) return "(-9223372036854775807LL - 1)";
    return std::to_string(v) + "LL";
}

uint64_t fnv(const std::string &s) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (unsigned char ch : s) h = (h ^ ch) * 0x100000001B3ull;
    return h;
}

// a second, unrelated 64-bit hash of the source: a cache entry must match both (and the length) before its code is loaded
uint64_t mix64(const std::string &s) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (unsigned char ch : s) {
        h += ch + 0x9E3779B97F4A7C15ull;
        h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
        h = (h ^ (h >> 27)) * 0x94D049BB133111EBull;
        h ^= h >> 31;
    }
    return h;
}

std::mutex g_mu;
std::map<std::string, std::vector<char>> g_code;        // key (hash + arch) -> code object, per process

// ---- the on-disk cache ---------------------------------------------------------------------------------------------------
// Code objects are device code that runs against this process's GPU memory, so a cache entry is only ever read from a
// directory that belongs to the caller and that nobody else can write: the directory (created 0700, parents included when
// missing) must be a real directory -- not a symbolic link -- owned by the effective user and closed to group and others;
// entries are regular files of the same owner opened with O_NOFOLLOW, and carry a header {magic, both hashes and the length
// of the source they were built from, code length} that must match the source about to be compiled.  The key holds the
// hiprtc version, so a toolchain upgrade does not meet stale code.  Anything that does not check out is ignored (and said on
// stderr once): the scan is compiled again, never loaded from there.
struct CacheHeader { char magic[8]; uint64_t h1, h2, src_len, code_len; };
const char kCacheMagic[8] = {'V', 'D', 'L', 'J', 'I', 'T', '2', 0};

bool make_private_dirs(const std::string &dir) {
    struct stat st;
    if (lstat(dir.c_str(), &st) == 0) return true;
    const size_t cut = dir.find_last_of('/');
    if (cut != std::string::npos && cut > 0 && !make_private_dirs(dir.substr(0, cut))) return false;
    return mkdir(dir.c_str(), 0700) == 0 || errno == EEXIST;
}
// "" when the directory cannot be trusted (why: on stderr, once per directory)
std::string trusted_cache_dir() {
    const char *dir = getenv("VDL_JIT_CACHE");
    if (!dir || !*dir) return "";
    static std::mutex mu;
    static std::map<std::string, bool> verdicts;
    std::lock_guard<std::mutex> g(mu);
    auto it = verdicts.find(dir);
    if (it != verdicts.end()) return it->second ? dir : "";
    std::string why;
    struct stat st;
    if (!make_private_dirs(dir)) why = "cannot be created";
    else if (lstat(dir, &st) != 0) why = "cannot be examined";
    else if (!S_ISDIR(st.st_mode)) why = "is not a directory (symbolic links are not followed)";
    else if (st.st_uid != geteuid()) why = "belongs to another user";
    else if (st.st_mode & (S_IWGRP | S_IWOTH)) why = "is writable by group or others";
    verdicts[dir] = why.empty();
    if (!why.empty()) fprintf(stderr, "vdl: VDL_JIT_CACHE=%s %s: not used, specialised scans are compiled in this process\n", dir, why.c_str());
    return why.empty() ? dir : "";
}
bool cache_read(const std::string &path, const std::string &src, std::vector<char> &code) {
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return false;
    bool ok = false;
    struct stat st;
    CacheHeader h;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid() && !(st.st_mode & (S_IWGRP | S_IWOTH)) &&
        read(fd, &h, sizeof h) == (ssize_t)sizeof h && memcmp(h.magic, kCacheMagic, 8) == 0 && h.h1 == fnv(src) && h.h2 == mix64(src) &&
        h.src_len == src.size() && h.code_len > 0 && (off_t)(sizeof h + h.code_len) == st.st_size) {
        code.resize(h.code_len);
        size_t got = 0;
        while (got < code.size()) {
            const ssize_t k = read(fd, code.data() + got, code.size() - got);
            if (k <= 0) break;
            got += (size_t)k;
        }
        ok = got == code.size();
    }
    close(fd);
    if (!ok) code.clear();
    return ok;
}
void cache_write(const std::string &path, const std::string &src, const std::vector<char> &code) {
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return;
    CacheHeader h;
    memcpy(h.magic, kCacheMagic, 8);
    h.h1 = fnv(src); h.h2 = mix64(src); h.src_len = src.size(); h.code_len = code.size();
    bool ok = write(fd, &h, sizeof h) == (ssize_t)sizeof h;
    size_t put = 0;
    while (ok && put < code.size()) {
        const ssize_t k = write(fd, code.data() + put, code.size() - put);
        if (k <= 0) ok = false; else put += (size_t)k;
    }
    ok = (close(fd) == 0) && ok;
    if (ok) ok = rename(tmp.c_str(), path.c_str()) == 0;
    if (!ok) unlink(tmp.c_str());
}

}  // namespace

// the descriptor as constexpr functions: only what differs from the defaults is written
static std::string desc_text(Kind kind, const MsArgs &C, const MScanDesc &D, const char *suffix = "") {
    std::ostringstream o;
    o << "namespace vdl {\n";
    o << "constexpr MsArgs jit_args" << suffix << "() {\n    MsArgs a{};\n";
    o << "    a.ncol = " << C.ncol << "; a.widths = " << C.widths << "ull; a.filtered = " << C.filtered << "u; a.derived = " << C.derived
      << "u; a.lazy = " << C.lazy << "u; a.stages = " << C.stages << "ull; a.queued = " << C.queued << ";\n    return a;\n}\n";
    o << "constexpr MScanDesc jit_desc" << suffix << "() {\n    MScanDesc d{};\n";
    o << "    d.nagg = " << D.nagg << "; d.nkey = " << D.nkey << "; d.replicas = " << D.replicas << "; d.pmin = " << lit(D.pmin) << "; d.pcount = " << lit(D.pcount) << ";\n";
    int pool = 0;
    for (int k = 0; k < C.ncol; k++) {
        if ((C.filtered >> k) & 1u) o << "    d.flo[" << k << "] = " << lit(D.flo[k]) << "; d.fhi[" << k << "] = " << lit(D.fhi[k]) << ";\n";
        if ((C.derived >> k) & 1u) {
            o << "    d.dkind[" << k << "] = " << D.dkind[k] << "; d.dsrc[" << k << "] = " << D.dsrc[k] << "; d.dsrc2[" << k << "] = " << D.dsrc2[k]
              << "; d.dtests[" << k << "] = " << D.dtests[k] << ";\n";
            if (D.dkind[k] == VC_FORM) pool = std::max(pool, D.dsrc[k] + D.dsrc2[k]);
        }
    }
    for (int f = 0; f < pool; f++)
        o << "    d.form[" << f << "].op = " << D.form[f].op << "; d.form[" << f << "].col = " << D.form[f].col << "; d.form[" << f << "].lo = " << lit(D.form[f].lo)
          << "; d.form[" << f << "].hi = " << lit(D.form[f].hi) << ";\n";
    o << "    d.ncomp = " << D.ncomp << "; d.key_masked = " << D.key_masked << "; d.key_mask = " << lit(D.key_mask) << ";\n";
    for (int k = 0; k < D.ncomp; k++)
        o << "    d.comp[" << k << "].col = " << D.comp[k].col << "; d.comp[" << k << "].rsh = " << D.comp[k].rsh << "; d.comp[" << k << "].lsh = " << D.comp[k].lsh
          << "; d.comp[" << k << "].sub = " << lit(D.comp[k].sub) << ";\n";
    for (int s = 0; s < D.nkey; s++)
        o << "    d.key[" << s << "].kind = (KeyStep::Kind)" << (int)D.key[s].kind << "; d.key[" << s << "].target = " << D.key[s].target << "; d.key[" << s << "].col = "
          << D.key[s].col << "; d.key[" << s << "].bin = " << D.key[s].bin << "; d.key[" << s << "].const_left = " << D.key[s].const_left << "; d.key[" << s
          << "].k = " << lit(D.key[s].k) << ";\n";
    for (int j = 0; j < D.nagg; j++) {
        const MAggDesc &a = D.agg[j];
        o << "    d.agg[" << j << "].kind = " << a.kind << "; d.agg[" << j << "].used = " << a.used << "u; d.agg[" << j << "].plain = " << a.plain << "u; d.agg[" << j
          << "].constant = " << lit(a.constant) << ";\n";
        for (int k = 0; k < C.ncol; k++)
            if ((a.used >> k) & 1u) o << "    d.agg[" << j << "].fa[" << k << "] = " << lit(a.fa[k]) << "; d.agg[" << j << "].fs[" << k << "] = " << lit(a.fs[k]) << ";\n";
    }
    if (kind != MSCAN) {
        o << "    d.take = " << D.take << "u; d.nout = " << D.nout << "; d.bitmap_only = " << D.bitmap_only << "; d.carry = " << D.carry << "u;\n";
        for (int k = 0; k < D.nout; k++) o << "    d.out_col[" << k << "] = " << D.out_col[k] << ";\n";
        o << "    d.nexpr = " << D.nexpr << ";\n";
        for (int k = 0; k < D.nexpr; k++) o << "    d.expr_at[" << k << "] = " << D.expr_at[k] << "; d.expr_len[" << k << "] = " << D.expr_len[k] << ";\n";
    }
    o << "    return d;\n}\n}  // namespace vdl\n";
    return o.str();
}

std::string scan_source(Kind kind, const MsArgs &C, const MScanDesc &D, const Shape &sh) {
    std::ostringstream o;
    o << "#define VDL_SPEC_UNROLL _Pragma(\"unroll\")\n";
    if (kind == MSCAN && sh.census) o << "#define VDL_CENSUS 1\n";
    if (kind == MSCAN && C.lazy && C.queued) {
        // the queue form: only the filters of the columns that come with the tile, as straight-line code over the tile's rows
        o << "#define VDL_QUEUE_FILTER";
        for (int c = 0; c < C.ncol; c++)
            if (!((C.derived >> c) & 1u) && !((C.lazy >> c) & 1u) && ((C.filtered >> c) & 1u))
                o << " _Pragma(\"unroll\") for (int r = 0; r < RW; r++) alive[r] = alive[r] & (v[" << c << "][r] >= " << lit(D.flo[c]) << ") & (v[" << c << "][r] <= " << lit(D.fhi[c]) << ");";
        o << "\n";
    } else if (kind == MSCAN && C.lazy) {
        // the stages of a scan that reads late (MsArgs::stages), as straight-line code inside the body's row loop context
        // a lane's rows come in adjacent pairs (2u, 2u + 1): with aligned columns a pair of which a row is still in is ONE
        // 16-byte (int64) / 8-byte (int32) load instead of two masked scalar ones; the one-row tail and unaligned columns
        // take the scalar form
        const bool pairs = sh.vec && !getenv("VDL_JIT_NO_PAIR_LOADS");
        // census builds: the wave's lanes hold consecutive addresses (stride 2 w), so a lane is the first asker of its line when no
        // lower active lane lies in the same 128 bytes -- the lanes [lane - (a % 128) / (2 w), lane)
        auto census = [&](int c, int w, const char *on) {
            std::ostringstream l;
            if (!sh.census) return std::string();
            l << " { const uint64_t m_ = __ballot(" << on << "); const uint64_t a_ = (uint64_t)Cr.ptr[" << c << "] + (uint64_t)(rowid[r] - Cr.row0) * " << w << "ull;"
              << " const int ln_ = (int)(threadIdx.x & 63u); int lo_ = ln_ - (int)((a_ & 127ull) / " << 2 * w << "ull); if (lo_ < 0) lo_ = 0;"
              << " const bool first_ = (" << on << ") && ((m_ >> lo_) & ((1ull << (ln_ - lo_)) - 1ull)) == 0ull;"
              << " const uint64_t f_ = __ballot(first_); if (ln_ == 0) census_cnt[" << c << "] += (unsigned long long)__popcll(f_); }";
            return l.str();
        };
        auto load = [&](int c, const char *mask) {
            std::ostringstream l;
            const int w = C.width(c);
            if (pairs && (w == 8 || w == 4)) {
                const char *vt = w == 8 ? "ll2" : "i32x2";
                l << " if (RW % 2 == 0) { _Pragma(\"unroll\") for (int r = 0; r < RW; r += 2) { const int r1 = r + 1 < RW ? r + 1 : r; v[" << c << "][r] = 0; v[" << c
                  << "][r1] = 0;" << census(c, w, (std::string(mask) + "[r] | " + mask + "[r1]").c_str()) << " if (" << mask << "[r] | " << mask << "[r1]) { const int64_t i0 = rowid[r] - Cr.row0; if (i0 + 1 < Cr.n) { const " << vt << " x = *(const " << vt
                  << " *)((const char *)Cr.ptr[" << c << "] + i0 * " << w << "); v[" << c << "][r] = x.x; v[" << c << "][r1] = x.y; } else if (" << mask << "[r]) v[" << c
                  << "][r] = load_scalar(Cr.ptr[" << c << "], " << w << ", i0); } } } else {";   /* (the table's last row has no partner) */
            }
            l << " _Pragma(\"unroll\") for (int r = 0; r < RW; r++) { v[" << c << "][r] = 0; if (" << mask << "[r]) v[" << c << "][r] = load_scalar(Cr.ptr[" << c << "], "
              << w << ", rowid[r] - Cr.row0); }";
            if (pairs && (w == 8 || w == 4)) l << " }";
            return l.str();
        };
        o << "#define VDL_STAGED_PRE";
        for (int st = 0; st <= 3; st++)
            for (int c = 0; c < C.ncol; c++) {
                if (((C.derived >> c) & 1u) || C.stage(c) != st) continue;
                if (st > 0) o << load(c, "alive");
                if ((C.filtered >> c) & 1u)
                    o << " _Pragma(\"unroll\") for (int r = 0; r < RW; r++) alive[r] = alive[r] & (v[" << c << "][r] >= " << lit(D.flo[c]) << ") & (v[" << c << "][r] <= "
                      << lit(D.fhi[c]) << ");";
            }
        for (int c = 0; c < C.ncol; c++) if (!((C.derived >> c) & 1u) && C.stage(c) == 14) o << load(c, "alive");
        o << "\n#define VDL_STAGED_POST";
        for (int c = 0; c < C.ncol; c++) if (!((C.derived >> c) & 1u) && C.stage(c) == 15) o << load(c, "pass");
        o << "\n";
    }
    o << "#define VDL_PROJ_U " << VDL_PROJ_U_HOST << "\n" << kEmbedded << "\n" << desc_text(kind, C, D);      // (the tile of the projection scan as this library was built)
    const char *b = sh.vec ? "true" : "false";
    if (kind == MSCAN)
        o << "extern \"C\" __global__ __launch_bounds__(256) void " << entry_name(kind, C, D, sh) << "(const vdl::MsArgs Cr, const vdl::MScanDesc *__restrict__ Dp) {\n"
             "    constexpr vdl::MsArgs C = vdl::jit_args();\n"
             "    constexpr vdl::MScanDesc D = vdl::jit_desc();\n"
             "    vdl::mscan_body<" << sh.nc << ", " << sh.u << ", " << b << ", " << b << ", " << (sh.grouped ? "true" : "false") << ", " << (sh.der ? "true" : "false")
          << ", " << (C.lazy ? "true" : "false") << ">(C, Cr, D, *Dp);\n}\n";
    else if (kind == SELECT)
        o << "extern \"C\" __global__ __launch_bounds__(256) void " << entry_name(kind) << "(const vdl::MsArgs Cr, const vdl::MScanDesc *__restrict__ Dp) {\n"
             "    constexpr vdl::MsArgs C = vdl::jit_args();\n"
             "    constexpr vdl::MScanDesc D = vdl::jit_desc();\n"
             "    vdl::project_select_body<" << sh.nc << ", vdl::kProjU, " << b << ", " << b << ">(C, Cr, D, *Dp);\n}\n";
    else
        o << "extern \"C\" __global__ __launch_bounds__(256) void " << entry_name(kind) << "(const vdl::MsArgs Cr, const vdl::MScanDesc *__restrict__ Dp, const uint16_t *__restrict__ scratch,\n"
             "        const int64_t *__restrict__ counts, const int64_t *__restrict__ offsets) {\n"
             "    constexpr vdl::MsArgs C = vdl::jit_args();\n"
             "    constexpr vdl::MScanDesc D = vdl::jit_desc();\n"
             "    vdl::project_take_body<" << sh.nc << ">(C, Cr, D, *Dp, scratch, counts, offsets);\n}\n";
    return o.str();
}
std::string mscan_source(const MsArgs &C, const MScanDesc &D, const Shape &sh) { return scan_source(MSCAN, C, D, sh); }
// the fused front in one pass: two descriptors -- the deciding columns as the select side numbers them, every column with the outputs
std::string front_source(const MsArgs &Cs, const MScanDesc &Ds, const MsArgs &Ct, const MScanDesc &Dt, const Shape &sh, int nct) {
    std::ostringstream o;
    o << "#define VDL_SPEC_UNROLL _Pragma(\"unroll\")\n";
    o << "#define VDL_PROJ_U " << VDL_PROJ_U_HOST << "\n#define VDL_FRONT_BATCH " << VDL_FRONT_BATCH_HOST << "\n" << kEmbedded << "\n" << desc_text(SELECT, Cs, Ds, "_s") << desc_text(TAKE, Ct, Dt, "_t");
    const char *b = sh.vec ? "true" : "false";
    // (Q3's front: 107 VGPRs = 4 blocks per CU; asked to fit 5 or 6 waves per SIMD -- 96 / 80 registers, 12 / 76 bytes of scratch -- the pass
    // took the same 2.64-2.66 ms at SF100: it moves 14.7 GB, i.e. 5.5 TB/s, and is bound by that, not by occupancy)
    o << "extern \"C\" __global__ __launch_bounds__(256) void " << entry_name(FRONT) << "(const vdl::MsArgs Csr, const vdl::MScanDesc *__restrict__ Dsp, const vdl::MsArgs Ctr,\n"
         "        const vdl::MScanDesc *__restrict__ Dtp, const vdl::FrontLook lk) {\n"
         "    constexpr vdl::MsArgs Cs = vdl::jit_args_s();\n"
         "    constexpr vdl::MScanDesc Ds = vdl::jit_desc_s();\n"
         "    constexpr vdl::MsArgs Ct = vdl::jit_args_t();\n"
         "    constexpr vdl::MScanDesc Dt = vdl::jit_desc_t();\n"
         "    vdl::project_front_body<" << sh.nc << ", " << nct << ", vdl::kProjU, " << b << ", " << b << ">(Cs, Csr, Ds, *Dsp, Ct, Ctr, Dt, *Dtp, lk);\n}\n";
    return o.str();
}
const char *entry_name(Kind kind) { return kind == MSCAN ? "vdl_jit_mscan" : kind == SELECT ? "vdl_jit_project_select" : kind == FRONT ? "vdl_jit_project_front" : "vdl_jit_project_take"; }
// aggregate scans carry their shape in the kernel's name, so that a profile of a tuned run lists the tuner's candidates apart
std::string entry_name(Kind kind, const MsArgs &C, const MScanDesc &D, const Shape &sh) {
    if (kind != MSCAN) return entry_name(kind);
    char tag[16];
    snprintf(tag, sizeof tag, "%06llx", (unsigned long long)(fnv(desc_text(kind, C, D)) & 0xffffffull));      // which plan's scan
    // (staged: how many filter columns come with the tile is part of the name too -- the profiles tell the forms apart by it)
    int eager_filters = 0;
    for (int c = 0; c < C.ncol; c++) eager_filters += ((C.filtered >> c) & 1u) && !((C.derived >> c) & 1u) && C.stage(c) == 0;
    return std::string("vdl_jit_mscan_") + (sh.grouped ? "grouped_" : "") + "u" + std::to_string(sh.u) + (C.lazy ? "_staged" + std::to_string(eager_filters) + "_" : "_") +
           (sh.census ? "census_" : "") + tag;
}

bool compile(const std::string &src, const std::string &arch, std::vector<char> &code, std::string &log) {
    std::string key = std::to_string(fnv(src)) + "_" + std::to_string(src.size()) + "_" + arch;
    {
        Rtc &r0 = rtc();
        int major = 0, minor = 0;
        if (r0.lib && r0.version && r0.version(&major, &minor) == 0) key += "_rtc" + std::to_string(major) + "." + std::to_string(minor);
    }
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto it = g_code.find(key);
        if (it != g_code.end()) { code = it->second; return true; }
    }
    if (const char *dump = getenv("VDL_JIT_DUMP")) {           // debugging: the generated translation unit as a file
        std::ofstream f(std::string(dump) + "/vdl_" + key + ".hip");
        f << src;
    }
    const std::string dir = trusted_cache_dir();
    std::string path;
    if (!dir.empty()) {
        path = dir + "/vdl_" + key + ".vdlco";
        if (cache_read(path, src, code)) { std::lock_guard<std::mutex> g(g_mu); g_code[key] = code; return true; }
    }
    Rtc &r = rtc();
    if (!r.lib || !r.why.empty()) { log = r.why.empty() ? "libhiprtc not usable" : r.why; return false; }
    void *prog = nullptr;
    if (r.create(&prog, src.c_str(), "vdl_jit_scan.hip", 0, nullptr, nullptr) != 0) { log = "hiprtcCreateProgram failed"; return false; }
    const std::string a = "--offload-arch=" + arch;
    const char *opts[] = {a.c_str(), "-O3", "-std=c++17"};
    const int rc = r.compile(prog, 3, opts);
    size_t n = 0;
    if (r.log_size(prog, &n) == 0 && n > 1) { log.resize(n); r.log(prog, &log[0]); }
    if (rc != 0) { if (log.empty()) log = "hiprtcCompileProgram failed (" + std::to_string(rc) + ")"; r.destroy(&prog); return false; }
    n = 0;
    r.code_size(prog, &n);
    code.resize(n);
    r.code(prog, code.data());
    r.destroy(&prog);
    if (code.empty()) { log = "hiprtc produced no code"; return false; }
    if (!path.empty()) cache_write(path, src, code);
    std::lock_guard<std::mutex> g(g_mu);
    g_code[key] = code;
    return true;
}

Kernel::~Kernel() { if (mod) (void)hipModuleUnload(mod); }

std::shared_ptr<Kernel> load(const std::vector<char> &code, std::string &why, const std::string &entry) {
    auto k = std::make_shared<Kernel>();
    hipError_t e = hipModuleLoadData(&k->mod, code.data());
    if (e != hipSuccess) { why = std::string("hipModuleLoadData: ") + hipGetErrorString(e); k->mod = nullptr; return nullptr; }
    e = hipModuleGetFunction(&k->fn, k->mod, entry.c_str());
    if (e != hipSuccess) { why = std::string("hipModuleGetFunction: ") + hipGetErrorString(e); return nullptr; }
    return k;
}

}  // namespace jit
}  // namespace vdl

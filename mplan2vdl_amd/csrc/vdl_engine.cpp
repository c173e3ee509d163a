// vdl_engine.cpp -- context, column catalog, HBM pool, plan execution and the C ABI (include/vdl.h).
//
// Two execution strategies, both HIP-only (there is no CPU fallback anywhere in this library):
//   * fused   : programs of the "filter -> gather -> global fold" shape run as one read-once scan
//               per table (vdl_fuse.cpp decides, k_scan executes);
//   * general : any other program runs statement by statement with one kernel per operator;
//               RangeV/RangeC, Project, Shuffle and identity Gathers never touch memory.
#include "vdl_genexec.h"

namespace {

// the descriptor `d` on the device under `role` (vdl_plan::desc_slots): uploaded when none of the role's copies already holds these
// bytes.  A role keeps up to three copies, the least recently used one is overwritten: the fused front's take descriptor names the
// output buffers, which alternate between two sets of addresses from run to run (the last run's results are still the plan's when
// the next run allocates), and with one copy per role every query paid a 5 us upload in the stream.
static const MScanDesc *desc_on_device(vdl_ctx *c, vdl_plan *p, const std::string &role, const MScanDesc &d) {
    constexpr int kCopies = 3;
    vdl_plan::DescSlot *hit = nullptr, *oldest = nullptr;
    for (int k = 0; k < kCopies; k++) {
        vdl_plan::DescSlot &sl = p->desc_slots[role + "#" + std::to_string(k)];
        if (sl.shadow.size() == sizeof(MScanDesc) && std::memcmp(sl.shadow.data(), &d, sizeof(MScanDesc)) == 0) { hit = &sl; break; }
        if (!oldest || sl.used < oldest->used) oldest = &sl;
    }
    uint64_t &clock = p->desc_clock;
    if (hit) { hit->used = ++clock; return (const MScanDesc *)hit->dev->p; }
    vdl_plan::DescSlot &sl = *oldest;
    if (!sl.dev) sl.dev = dev_alloc(c, sizeof(MScanDesc));
    // the shadow is the SOURCE of the copy: it stays put until the next upload, the caller's `d` may be a local
    sl.shadow.assign((const unsigned char *)&d, (const unsigned char *)&d + sizeof(MScanDesc));
    sl.used = ++clock;
    HIP_CHECK(hipMemcpyAsync(sl.dev->p, sl.shadow.data(), sizeof(MScanDesc), hipMemcpyHostToDevice, c->stream));
    return (const MScanDesc *)sl.dev->p;
}

// ------------------------------------------------------------------------------------------------
// fused execution
// ------------------------------------------------------------------------------------------------
struct NeedGeneralPath : Error {          // a fused assumption did not hold for this data: rerun unfused
    explicit NeedGeneralPath(const std::string &m) : Error(VDL_ERR_UNSUPPORTED, m) {}
};

int64_t plan_words(const vdl_plan *p, std::vector<int32_t> *ops, bool *shardable) {
    int64_t off = 0;
    if (shardable) *shardable = true;
    for (const ScanPlan &sp : p->fused.scans) {
        if (ops) {
            ops->push_back(VDL_REDUCE_SUM);
            for (const ScanAgg &ag : sp.aggs) ops->push_back(ag.kind == AGG_SUM ? VDL_REDUCE_SUM : ag.kind == AGG_MIN ? VDL_REDUCE_MIN : VDL_REDUCE_MAX);
        }
        off += (int64_t)sp.aggs.size() + 1;
    }
    for (const GroupScanPlan &gp : p->fused.gscans) {
        for (int64_t b = 0; b < gp.pcount; b++) {
            if (ops) ops->push_back(VDL_REDUCE_SUM);
            for (const ScanAgg &ag : gp.aggs) {
                if (ops) ops->push_back(ag.kind == AGG_SUM ? VDL_REDUCE_SUM : ag.kind == AGG_MAX ? VDL_REDUCE_MAX
                                        : ag.kind == AGG_FIRST ? VDL_REDUCE_FIRST : VDL_REDUCE_MIN);
            }
        }
        if (ops) ops->push_back(VDL_REDUCE_SUM);                              // out-of-domain key count
        off += gp.pcount * ((int64_t)gp.aggs.size() + 1) + 1;
    }
    // a semi-join set is built from ALL selected rows of its source table: a rank that holds a shard of either table would test
    // (or build) a partial set
    if (shardable) for (const PreludeItem &it : p->fused.prelude) if (it.kind == PreludeItem::SEMI_BITMAP) *shardable = false;
    return off;
}

// single-aggregate scans over <= 4 columns take the tuned k_scan; everything else k_mscan
static bool use_kscan(const ScanPlan &sp) {
    for (const ScanColumn &c : sp.cols) if (c.kind != VC_DIRECT) return false;       // derived columns (fused join scans): k_mscan
    if (getenv("VDL_NO_KSCAN")) return false;                                         // experiments: everything through k_mscan
    return sp.aggs.size() == 1 && sp.cols.size() <= 4;
}

// formula columns (VC_FORM) into the descriptor's pool, in the kernels' layout (MScanDesc::form): the range tests sorted by
// column, then the postfix program with every test replaced by a reference to its result bit
static void bind_forms(const std::vector<ScanColumn> &sc, MScanDesc &d) {
    int used = 0;
    for (size_t k = 0; k < sc.size(); k++) {
        if (sc[k].kind != VC_FORM) continue;
        const std::vector<FormStep> &prog = sc[k].form;
        std::vector<int> tests;
        for (size_t i = 0; i < prog.size(); i++) if (prog[i].op == FormStep::LEAF) tests.push_back((int)i);
        std::stable_sort(tests.begin(), tests.end(), [&](int x, int y) { return prog[(size_t)x].col < prog[(size_t)y].col; });
        if (tests.size() > 64 || used + (int)(tests.size() + prog.size()) > kMaxFormPool)
            throw Error(VDL_ERR_UNSUPPORTED, "the scan's conditions do not fit the descriptor (" + std::to_string(kMaxFormPool) + " steps in all, 64 tests per condition)");
        std::vector<int> bit_of(prog.size(), -1);
        d.dsrc[k] = used;
        d.dtests[k] = (int)tests.size();
        for (size_t t = 0; t < tests.size(); t++) { bit_of[(size_t)tests[t]] = (int)t; d.form[used++] = prog[(size_t)tests[t]]; }
        for (size_t i = 0; i < prog.size(); i++) {
            FormStep f = prog[i];
            if (f.op == FormStep::LEAF) { f.op = FormStep::REF; f.col = bit_of[i]; f.lo = f.hi = 0; }
            d.form[used++] = f;
        }
        d.dsrc2[k] = used - d.dsrc[k];
    }
}

// A row-id column that only CONDITIONS read (the reference's anti-join shape `k - row id` as a truth value: every row but row k)
// means the row's number in the table: on a rank of a sharded run it counts from the table's first row.  One that INDEXES a lookup
// (the bit of the row's own id in a set built over this shard's rows: co-partitioned placements) keeps counting from the shard's.
static bool rowid_counts_from_the_table(const std::vector<ScanColumn> &sc) {
    bool any = false, as_index = false;
    for (const ScanColumn &c : sc) any |= c.kind == VC_ROWID;
    for (const ScanColumn &c : sc) {
        if (c.kind == VC_DIRECT || c.kind == VC_FORM || c.kind == VC_ROWID) continue;
        for (int src : {c.idx, c.idx2}) if (src >= 0 && (size_t)src < sc.size() && sc[(size_t)src].kind == VC_ROWID) as_index = true;
    }
    return any && !as_index;
}

template <typename PlanT>
int64_t bind_mscan(vdl_ctx *c, const PlanT &sp, MScanCols &cols, MScanDesc &d, int64_t *bytes_per_row, int64_t row0) {
    cols = MScanCols{};
    d = MScanDesc{};
    cols.ncol = (int)sp.cols.size();
    d.nagg = (int)sp.aggs.size();
    int64_t n = -1;
    *bytes_per_row = 0;
    for (int k = 0; k < cols.ncol; k++) {
        const ScanColumn &sc = sp.cols[(size_t)k];
        cols.kind[k] = sc.kind;
        cols.lo[k] = sc.lo; cols.hi[k] = sc.hi;
        cols.filtered[k] = (cols.lo[k] != INT64_MIN || cols.hi[k] != INT64_MAX) ? 1 : 0;
        d.flo[k] = cols.lo[k]; d.fhi[k] = cols.hi[k];
        d.dkind[k] = sc.kind; d.dsrc[k] = sc.idx; d.dsrc2[k] = sc.idx2;
        if (sc.kind == VC_DIRECT) {
            const Column &col = find_col(c, sc.name);
            if (n >= 0 && col.n != n)
                throw Error(VDL_ERR_SHAPE, "columns of table '" + sp.table + "' have different lengths in the catalog");
            n = col.n;
            cols.ptr[k] = col.dev; cols.width[k] = col.width;
            *bytes_per_row += col.width;
        } else if (sc.kind == VC_GATHER || sc.kind == VC_INRANGE) {      // a column of another table, looked up / its length
            const Column &col = find_col(c, sc.name);
            cols.ptr[k] = col.dev; cols.width[k] = col.width;
            d.dn[k] = col.n;
        } else {                                                          // VC_BITS / VC_LUT: filled in by run_prelude before every launch
            cols.ptr[k] = nullptr; cols.width[k] = 8;
            d.dn[k] = 0;
        }
    }
    bind_forms(sp.cols, d);
    cols.n = n;
    cols.row0 = row0;
    cols.rowid_global = rowid_counts_from_the_table(sp.cols);
    for (int j = 0; j < d.nagg; j++) {
        const ScanAgg &ag = sp.aggs[(size_t)j];
        MAggDesc &m = d.agg[j];
        m.kind = ag.kind;
        m.constant = ag.constant;
        for (const ScanFactor &f : ag.fac) {
            m.used |= 1u << f.col;
            if (f.a == 0 && f.s == 1) m.plain |= 1u << f.col;
            m.fa[f.col] = f.a; m.fs[f.col] = f.s;
        }
    }
    return n;
}

// Run-time specialisation of one multi-aggregate scan (vdl_jit.cpp): the shape of the precompiled variant the launch
// configuration chose, with exactly this scan's column count, and the descriptor as constants.  On success the kernel, its
// grid (occupancy of the specialised code) and name replace the variant's; on failure the variant stays and the note says why.
static jit::Shape jit_shape(const MScanCols &cols, const ScanLaunch &cfg) {
    jit::Shape sh;
    mscan_variant_shape(cfg, &sh.nc, &sh.u, &sh.vec, &sh.grouped, &sh.der);
    sh.nc = cols.ncol;
    for (int k = 0; k < cols.ncol; k++) sh.der |= cols.kind[k] != VC_DIRECT;
    const char *u = getenv(sh.grouped ? "VDL_JIT_GROUP_U" : "VDL_JIT_U");
    if (u && atoi(u) >= 1 && atoi(u) <= 8) sh.u = atoi(u);
    return sh;
}
static std::string jit_name(const jit::Shape &sh) {
    return "k_mscan_specialised<" + std::to_string(sh.nc) + "," + std::to_string(sh.u) + "," + (sh.vec ? "vec" : "novec") + "," + (sh.grouped ? "grouped" : "global") +
           (sh.der ? ",derived" : "") + ">";
}
struct Specialised { std::shared_ptr<jit::Kernel> k; int grid = 0, per_cu = 0, u = 0, lazy = 0; size_t code_bytes = 0; std::string name, stages; };
// "l_discount@1 l_quantity@2 l_extendedprice@last": which table columns a staged scan reads when (MsArgs::stages)
static std::string stages_text(const vdl_plan *p, size_t s, const MsArgs &args) {
    const size_t ns = p->fused.scans.size();
    const std::vector<ScanColumn> &sc = s < ns ? p->fused.scans[s].cols : p->fused.gscans[s - ns].cols;
    std::string o;
    for (int k = 0; k < args.ncol && k < (int)sc.size(); k++) {
        const int st = args.stage(k);
        if (!st) continue;
        const std::string &name = sc[(size_t)k].name;
        o += (o.empty() ? "" : " ") + name.substr(name.find('.') == std::string::npos ? 0 : name.find('.') + 1) + "@" +
             (st == 15 ? std::string("last") : st == 14 ? std::string("lookups") : std::to_string(st));
    }
    return o;
}
// fraction of a table column's rows inside [lo, hi], from 16 samples of 4096 rows spread over the column
static double sampled_selectivity(vdl_ctx *c, const void *dev, int width, int64_t n, int64_t lo, int64_t hi) {
    if (const char *a = getenv("VDL_JIT_ASSUME_SELECTIVITY")) return atof(a);      // (tests without a GPU: vdl_plan_jit_check of staged builds)
    if (n <= 0 || !dev || c->device < 0) return 1.0;
    const int64_t chunk = std::min<int64_t>(4096, n), pieces = std::max<int64_t>(1, std::min<int64_t>(16, n / chunk));
    std::vector<char> host((size_t)(chunk * width));
    int64_t in = 0, seen = 0;
    for (int64_t k = 0; k < pieces; k++) {
        const int64_t at = pieces > 1 ? (n - chunk) / (pieces - 1) * k : 0;
        if (hipMemcpy(host.data(), (const char *)dev + at * width, (size_t)(chunk * width), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return 1.0; }
        for (int64_t i = 0; i < chunk; i++) {
            const int64_t x = width == 8 ? ((const int64_t *)host.data())[i] : width == 4 ? ((const int32_t *)host.data())[i]
                            : width == 2 ? ((const int16_t *)host.data())[i] : ((const int8_t *)host.data())[i];
            in += x >= lo && x <= hi;
        }
        seen += chunk;
    }
    return seen ? (double)in / (double)seen : 1.0;
}

// Stages of a specialised scan that reads late (MsArgs::stages): the most selective filter column on table columns comes
// with the tile, the other filter columns in order of (sampled) selectivity for the rows still in, then the sources of
// derived columns and of the group key, and last the columns that are only aggregate inputs.  0 = nothing to defer.
static uint64_t staged_columns(vdl_ctx *c, const MScanCols &cols, const MScanDesc &d, bool grouped, uint32_t *lazy_mask, int eager_filters = 1) {
    uint32_t source = 0, used = 0;
    for (int k = 0; k < cols.ncol; k++) {
        if (cols.kind[k] == VC_DIRECT) continue;
        if (cols.kind[k] == VC_FORM) { for (int f = d.dsrc[k]; f < d.dsrc[k] + d.dtests[k]; f++) source |= 1u << d.form[f].col; continue; }
        if (d.dsrc[k] >= 0) source |= 1u << d.dsrc[k];
        if (d.dsrc2[k] >= 0) source |= 1u << d.dsrc2[k];
    }
    if (grouped) {
        for (int k = 0; k < d.nkey; k++) if (d.key[k].kind == KeyStep::LOAD) source |= 1u << d.key[k].col;
        for (int k = 0; k < d.ncomp; k++) source |= 1u << d.comp[k].col;
    }
    for (int j = 0; j < d.nagg; j++) if (d.agg[j].kind != AGG_FIRST) used |= d.agg[j].used;
    std::vector<std::pair<double, int>> filters;
    for (int k = 0; k < cols.ncol; k++)
        if (cols.kind[k] == VC_DIRECT && cols.filtered[k])
            filters.push_back({sampled_selectivity(c, cols.ptr[k], cols.width[k], cols.n, cols.lo[k], cols.hi[k]), k});
    std::sort(filters.begin(), filters.end());
    uint64_t stages = 0;
    uint32_t lazy = 0;
    auto put = [&](int k, int st) { stages |= (uint64_t)st << (4 * k); if (st) lazy |= 1u << k; };
    const bool selective = !filters.empty() && filters[0].first < 0.6;
    // (eager_filters = 2: the second most selective filter column comes with the tile as well -- when the first leaves 14 % of
    // the rows, 71 % of the second's sectors are touched anyway and 16-byte streaming loads beat masked 8-byte ones)
    if (selective)
        for (size_t i = (size_t)std::max(eager_filters, 1); i < filters.size(); i++) put(filters[i].second, (int)std::min<size_t>(i - (size_t)std::max(eager_filters, 1) + 1, 3));
    for (int k = 0; k < cols.ncol; k++) {
        if (cols.kind[k] != VC_DIRECT || cols.filtered[k]) continue;
        if ((source >> k) & 1u) { if (selective) put(k, 14); }
        else if ((used >> k) & 1u) put(k, 15);
    }
    *lazy_mask = lazy;
    return stages;
}
static bool build_specialised(vdl_ctx *c, vdl_plan *p, size_t s, bool grouped, int u, int lazy /* 0 | eager filter columns of the staged form */, Specialised &out, std::string &why,
                              bool census = false) {
    jit::Shape sh = jit_shape(p->mcols[s], p->mcfg[s]);
    if (u > 0) sh.u = u;
    sh.census = census;
    std::vector<char> code;
    MsArgs args = mscan_args(p->mcols[s]);
    if (lazy) {
        args.stages = staged_columns(c, p->mcols[s], p->mdesc[s], grouped, &args.lazy, lazy == 3 ? 1 : lazy);
        if (!args.lazy) { why = "no column to read late"; return false; }
        if (lazy == 3) {
            // the queue form: the most selective filter column with the tile, EVERY other table column for the queued rows
            int eager = 0;
            for (int k = 0; k < p->mcols[s].ncol; k++) {
                if (p->mcols[s].kind[k] != VC_DIRECT) continue;
                if (!((args.lazy >> k) & 1u)) { eager++; if (!p->mcols[s].filtered[k]) { why = "a column that is no filter would come with the tile"; return false; } }
            }
            if (eager != 1) { why = "the queue form wants exactly one filter column with the tile"; return false; }
            args.queued = 1;
            args.stages = 0;
        }
    }
    if (!jit::compile(jit::mscan_source(args, p->mdesc[s], sh), c->arch, code, why)) { why = why.substr(0, 400); return false; }
    // a specialised scan is 10-25 KB of code; ten times that means the compiler did not fold the descriptor (it then sits in
    // scratch memory and every descriptor-driven loop stays): such a build is slower than the precompiled kernel
    if (code.size() > (size_t)96 << 10) { why = "the descriptor did not fold (" + std::to_string(code.size()) + " B of code)"; return false; }
    out.k = jit::load(code, why, jit::entry_name(jit::MSCAN, args, p->mdesc[s], sh));
    if (!out.k) return false;
    int per_cu = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, out.k->fn, 256, mscan_lds_bytes(p->mdesc[s], grouped)) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 2;
    }
    if (per_cu > 8) per_cu = 8;
    const int64_t tile = (int64_t)256 * 2 * sh.u;
    int64_t grid = (int64_t)c->num_cus * per_cu;
    if (grid > p->mcols[s].n / tile) grid = p->mcols[s].n / tile;
    if (grid < 1) grid = 1;
    out.grid = (int)grid; out.per_cu = per_cu; out.code_bytes = code.size(); out.name = jit_name(sh); out.u = sh.u; out.lazy = lazy;
    if (lazy) out.name.insert(out.name.size() - 1, lazy == 3 ? ",queue" : lazy > 1 ? ",late2" : ",late");
    if (lazy) out.stages = stages_text(p, s, args);
    return true;
}
static bool specialise_scan(vdl_ctx *c, vdl_plan *p, size_t s, bool grouped, std::string *kname) {
    Specialised sp;
    std::string why;
    // (tests, profiles: VDL_JIT_LATE=1|2 forces the staged form -- with that many filter columns read with the tile -- where a column allows it)
    const int late = getenv("VDL_JIT_LATE") ? std::max(1, atoi(getenv("VDL_JIT_LATE"))) : 0;
    if (!(late && build_specialised(c, p, s, grouped, 0, late, sp, why)) && !build_specialised(c, p, s, grouped, 0, 0, sp, why)) { p->jit_note += "scan " + std::to_string(s) + ": not specialised (" + why + "); "; return false; }
    p->mcfg[s].grid = sp.grid;
    p->mjit[s] = sp.k;
    p->mjit_form[s].u = sp.u; p->mjit_form[s].lazy = sp.lazy;
    *kname = sp.name;
    p->jit_note += "scan " + std::to_string(s) + ": " + sp.name + ", " + std::to_string(sp.code_bytes) + " B of code, " + std::to_string(sp.per_cu) + " blocks/CU" +
                   (sp.stages.empty() ? "" : ", read late: " + sp.stages) + "; ";
    return true;
}
// blocks a specialised scan may be launched with, whatever rows-per-lane the tuner settles on: the partials area is sized for it
static int max_scan_grid(const vdl_ctx *c, const vdl_plan *p, int chosen) { return p->use_jit ? std::max(chosen, c->num_cus * 8) : chosen; }

// vdl_plan_set_jit(plan, 2): at the first run, with the real columns and lookup tables in place, every specialised scan is
// built in up to eleven forms (a second each) -- 2, 3, 4, 6 row pairs per lane, then the staged forms that read late (one or two
// filter columns with the tile) at the winner's and at smaller shapes -- and the quickest of three timed launches stays; for a
// single-aggregate scan the hand-tuned k_scan is timed as well.  Which one wins depends on the registers the specialised
// code needs, on how its blocks fill the CUs and on the filters' selectivity: Q1 at SF100 measured 4.13 / 4.04 / 4.30 /
// 3.96 ms for 2 / 3 / 4 / 6 pairs (staged: 4.1-4.2), Q6 2.7 / 2.5 / 2.4 / 2.5 ms eager, 1.63 staged, 2.38 on k_scan.
static void tune_specialised(vdl_ctx *c, vdl_plan *p, int64_t *dev_words) {
    const size_t ns = p->fused.scans.size(), ng = p->fused.gscans.size();
    struct Events {                                            // (destroyed on every way out, also a throwing HIP_CHECK)
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    HIP_CHECK(hipEventCreate(&ev.a));
    HIP_CHECK(hipEventCreate(&ev.b));
    const hipEvent_t e0 = ev.a, e1 = ev.b;
    for (size_t s = 0; s < ns + ng; s++) {
        if (!p->mjit[s]) continue;
        const bool grouped = s >= ns;
        int64_t *out = dev_words + (grouped ? p->gword_offset[s - ns] : p->word_offset[s]);
        Specialised best;
        float best_ms = 0;
        std::string tried;
        // rows per lane first; then, at the winner, at 2 and at 1, the staged form that reads late (fewer rows per lane suit it:
        // its loads depend on each other, and what hides them is more waves, not more loads per wave)
        // (3 = the queue form: one filter column with the tile, the rows still in queued per wave and finished 64 at a time)
        std::vector<std::pair<int, int>> cands = {{2, 0}, {3, 0}, {4, 0}, {6, 0}, {0, 1}, {3, 1}, {2, 1}, {1, 1}, {3, 2}, {2, 2}, {4, 2}, {4, 3}, {3, 3}, {6, 3}};
        // VDL_JIT_PIN="u=3,late=2" (profiles: tools/profile_bench.sh runs the form a plain run chose, and nothing else): one candidate
        int pin_u = 0, pin_late = -1;
        if (const char *pin = getenv("VDL_JIT_PIN")) {
            if (const char *q = strstr(pin, "u=")) pin_u = atoi(q + 2);
            if (const char *q = strstr(pin, "late=")) pin_late = atoi(q + 5);
            if (pin_u > 0) cands = {{pin_u, std::max(pin_late, 0)}};
        }
        // a candidate's time is the MEDIAN of five launches after the module's first, and a later candidate only replaces the one
        // in hand when it is more than 2 % quicker: forms within the noise of each other no longer swap places from run to run
        auto median_of = [](std::vector<float> &t) { std::sort(t.begin(), t.end()); return t[t.size() / 2]; };
        int best_u = 0;
        for (auto &cu : cands) {
            const int u = cu.first ? cu.first : best_u;
            const int lazy = cu.second;
            if (u <= 0 || (int64_t)256 * 2 * u > p->mcols[s].n) continue;
            if (lazy == 1 && cu.first == best_u) continue;
            Specialised cand;
            std::string why;
            if (!build_specialised(c, p, s, grouped, u, lazy, cand, why)) continue;
            ScanLaunch cfg = p->mcfg[s];
            cfg.grid = cand.grid;
            HIP_CHECK(hipMemcpyAsync(p->mdev[s]->p, &p->mdesc[s], sizeof(MScanDesc), hipMemcpyHostToDevice, c->stream));
            std::vector<float> times;
            for (int rep = 0; rep < 6; rep++) {
                HIP_CHECK(hipEventRecord(e0, c->stream));
                HIP_CHECK(launch_mscan(p->mcols[s], p->mdesc[s], (const MScanDesc *)p->mdev[s]->p, cfg, grouped, false, out, false, c->stream, cand.k->fn));
                HIP_CHECK(hipEventRecord(e1, c->stream));
                HIP_CHECK(hipEventSynchronize(e1));
                float t = 0;
                HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
                if (rep > 0) times.push_back(t);                // the first launch of a module pays for its load
            }
            const float ms = median_of(times);
            tried += " u=" + std::to_string(u) + (lazy == 3 ? ",queue:" : lazy > 1 ? ",late2:" : lazy ? ",late:" : ":") + std::to_string((int)(ms * 1000)) + "us";
            if (!best.k || ms < best_ms * 0.98f) { best = cand; best_ms = ms; }
            if (!lazy && (best_u == 0 || cand.k == best.k)) best_u = u;       // the staged forms start from the quickest eager shape
        }
        if (!best.k) continue;
        if (!grouped && use_kscan(p->fused.scans[s]) && p->block_partials[s] && pin_u <= 0) {
            // the hand-tuned single-aggregate kernel is a candidate too
            float ms = 1e30f;
            std::vector<float> times;
            for (int rep = 0; rep < 6; rep++) {
                HIP_CHECK(hipEventRecord(e0, c->stream));
                HIP_CHECK(launch_scan(p->sargs[s], p->scfg[s], c->stream));
                HIP_CHECK(launch_scan_finish(p->sargs[s].block_partials, p->scfg[s].grid, p->sargs[s].nagg, nullptr, p->sargs[s], out, c->stream));
                HIP_CHECK(hipEventRecord(e1, c->stream));
                HIP_CHECK(hipEventSynchronize(e1));
                float t = 0;
                HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
                if (rep > 0) times.push_back(t);
            }
            ms = median_of(times);
            tried += std::string(" k_scan:") + std::to_string((int)(ms * 1000)) + "us";
            if (ms < best_ms * 0.98f) {
                p->kscan[s] = 1;
                p->mjit[s] = nullptr;
                p->jit_note += "scan " + std::to_string(s) + " tuned:" + tried + " -> " + scan_kernel_name(p->scfg[s]) + "; ";
                if ((int)s == p->dominant) p->dominant_kernel = std::string(scan_kernel_name(p->scfg[s])) + "_grid" + std::to_string(p->scfg[s].grid);
                continue;
            }
        }
        p->mjit[s] = best.k;
        p->mcfg[s].grid = best.grid;
        p->mjit_form[s].u = best.u; p->mjit_form[s].lazy = best.lazy;
        p->jit_note += "scan " + std::to_string(s) + " tuned:" + tried + " -> " + best.name + (best.stages.empty() ? "" : " (read late: " + best.stages + ")") + "; ";
        if ((int)s == p->dominant)
            p->dominant_kernel = best.name + "_grid" + std::to_string(best.grid) + (grouped ? "_rep" + std::to_string(p->mdesc[s].replicas) : "");
    }
    p->description = describe_plan(p);
}

// HBM bytes one launch of the dominant scan moves, counted rather than modelled.  The memory side fetches whole 128-byte lines,
// one request per line, whatever part of the line the lanes ask for (tools/ubench/fetch_calib: TCC_EA0_RDREQ = lines touched
// for streaming, every-other-sector and random masked 16-byte loads alike; FETCH_SIZE = 64 B per request).  A scan that reads
// every column with the tile moves its algorithmic bytes.  A staged scan (late materialisation) moves the eager columns in
// full plus, per late column, 128 B for every line in which some row was still in when the column was read: a CENSUS build of
// the very form that ran (same rows per lane, same stages; vdl_jit.cpp VDL_CENSUS) counts those lines in one untimed launch
// over the real columns.  detail: "column=bytes ..." for the note.
static int64_t scan_bytes_moved(vdl_ctx *c, vdl_plan *p, std::string &detail) {
    if (!p->bound || p->dominant < 0) throw Error(VDL_ERR_ARG, "vdl_plan_scan_traffic: run the (fused) plan first");
    const size_t s = (size_t)p->dominant, ns = p->fused.scans.size();
    const bool grouped = s >= ns;
    const std::vector<ScanColumn> &sc = grouped ? p->fused.gscans[s - ns].cols : p->fused.scans[s].cols;
    const MScanCols &cols = p->mcols[s];
    auto short_name = [&](int k) { const std::string &n = sc[(size_t)k].name; return n.substr(n.find('.') == std::string::npos ? 0 : n.find('.') + 1); };
    const bool staged = !(!grouped && p->kscan[s]) && p->mjit[s] && p->mjit_form[s].lazy > 0;
    int64_t total = 0;
    if (!grouped && p->kscan[s]) {                             // the hand-tuned single-aggregate kernel: its own argument block
        const ScanArgs &a = p->sargs[s];
        for (int k = 0; k < a.ncol; k++) { total += a.n * a.width[k]; detail += short_name(k) + "=" + std::to_string(a.n * a.width[k]) + " "; }
        detail += "(every column read with the tile)";
        return total;
    }
    if (!staged) {
        for (int k = 0; k < cols.ncol; k++)
            if (cols.kind[k] == VC_DIRECT) { total += cols.n * cols.width[k]; detail += short_name(k) + "=" + std::to_string(cols.n * cols.width[k]) + " "; }
        detail += "(every column read with the tile)";
        return total;
    }
    Specialised cen;
    std::string why;
    if (!build_specialised(c, p, s, grouped, p->mjit_form[s].u, p->mjit_form[s].lazy, cen, why, true)) throw Error(VDL_ERR_UNSUPPORTED, "the census build of the staged scan failed: " + why);
    BufP counts = dev_alloc(c, sizeof(unsigned long long) * kMaxVCols);
    BufP words = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(p->n_words, 1));
    HIP_CHECK(hipMemsetAsync(counts->p, 0, sizeof(unsigned long long) * kMaxVCols, c->stream));
    MScanDesc d = p->mdesc[s];
    d.census = (unsigned long long *)counts->p;
    BufP ddev = dev_alloc(c, sizeof(MScanDesc));
    HIP_CHECK(hipMemcpyAsync(ddev->p, &d, sizeof d, hipMemcpyHostToDevice, c->stream));
    ScanLaunch cfg = p->mcfg[s];
    cfg.grid = cen.grid;
    int64_t *out = (int64_t *)words->p + (grouped ? p->gword_offset[s - ns] : p->word_offset[s]);
    HIP_CHECK(launch_mscan(cols, d, (const MScanDesc *)ddev->p, cfg, grouped, false, out, false, c->stream, cen.k->fn));
    unsigned long long lines[kMaxVCols] = {};
    c->fetch_to_host(counts->p, kMaxVCols, (int64_t *)lines, c->stream);
    const MsArgs args = [&] { MsArgs a = mscan_args(cols); uint32_t lz = 0; a.stages = staged_columns(c, cols, p->mdesc[s], grouped, &lz, p->mjit_form[s].lazy == 3 ? 1 : p->mjit_form[s].lazy); a.lazy = lz; return a; }();
    for (int k = 0; k < cols.ncol; k++) {
        if (cols.kind[k] != VC_DIRECT) continue;
        const bool late = (args.lazy >> k) & 1u;
        const int64_t b = late ? (int64_t)lines[k] * 128 : cols.n * cols.width[k];
        total += b;
        detail += short_name(k) + "=" + std::to_string(b) + (late ? "(late: " + std::to_string(lines[k]) + " lines of " + std::to_string((cols.n * cols.width[k] + 127) / 128) + ") " : " ");
    }
    detail += "(census of " + cen.name + ", full tiles)";
    return total;
}

void bind_fused(vdl_ctx *c, vdl_plan *p) {
    const FusedPlan &F = p->fused;
    const size_t ns = F.scans.size(), ng = F.gscans.size();
    p->sargs.assign(ns, ScanArgs{});
    p->scfg.assign(ns, ScanLaunch{});
    p->block_partials.assign(ns, nullptr);
    p->word_offset.assign(ns, 0);
    p->mcols.assign(ns + ng, MScanCols{});
    p->mdesc.assign(ns + ng, MScanDesc{});
    p->mcfg.assign(ns + ng, ScanLaunch{});
    p->mparts.assign(ns + ng, nullptr);
    p->mdev.resize(ns + ng);
    p->mjit.assign(ns + ng, nullptr);
    p->mjit_form.assign(ns + ng, vdl_plan::JitForm{});
    p->kscan.assign(ns + ng, 0);
    p->jit_note.clear();
    p->jit_tuned = false;
    p->gword_offset.assign(ng, 0);
    p->reduce_ops.clear();
    bool shardable = true;
    p->n_words = plan_words(p, &p->reduce_ops, &shardable);
    int64_t off = 0;
    p->scan_rows = 0; p->scan_bytes = 0; p->dominant = -1;
    p->dominant_kernel = "none";
    for (size_t s = 0; s < ns; s++) {
        const ScanPlan &sp = F.scans[s];
        int64_t bpr = 0, n = 0;
        std::string kname;
        const bool eligible = use_kscan(sp);
        p->kscan[s] = eligible && !(p->use_jit && !sp.never);    // specialising: the single-aggregate scan runs through the general body too
        if (eligible) {                                       // (bound either way: the tuner compares the two)
            ScanArgs &a = p->sargs[s];
            a.ncol = (int)sp.cols.size(); a.nagg = 1; a.never = sp.never ? 1 : 0;
            n = -1;
            for (int k = 0; k < a.ncol; k++) {
                const Column &col = find_col(c, sp.cols[(size_t)k].name);
                if (n >= 0 && col.n != n)
                    throw Error(VDL_ERR_SHAPE, "columns of table '" + sp.table + "' have different lengths in the catalog");
                n = col.n;
                a.ptr[k] = col.dev; a.width[k] = col.width;
                a.lo[k] = sp.cols[(size_t)k].lo; a.hi[k] = sp.cols[(size_t)k].hi;
                a.filtered[k] = (a.lo[k] != INT64_MIN || a.hi[k] != INT64_MAX) ? 1 : 0;
                bpr += col.width;
            }
            a.n = n;
            const ScanAgg &ag = sp.aggs[0];
            a.kind[0] = ag.kind; a.constant[0] = ag.constant;
            for (const ScanFactor &f : ag.fac) {
                a.used[0] |= 1u << f.col;
                if (f.a == 0 && f.s == 1) a.plain[0] |= 1u << f.col;
                a.fa[0][f.col] = f.a; a.fs[0][f.col] = f.s;
            }
            p->scfg[s] = scan_launch_config(a, c->num_cus);
            if (p->scfg[s].variant < 0) throw Error(VDL_ERR_UNSUPPORTED, "no scan kernel variant for this shape");
            p->block_partials[s] = dev_alloc(c, sizeof(int64_t) * (size_t)p->scfg[s].grid * 2);
            a.block_partials = (int64_t *)p->block_partials[s]->p;
            kname = std::string(scan_kernel_name(p->scfg[s])) + "_grid" + std::to_string(p->scfg[s].grid);
        }
        if (!p->kscan[s]) {
            bpr = 0;
            n = bind_mscan(c, sp, p->mcols[s], p->mdesc[s], &bpr, p->row_offset);
            p->mcfg[s] = mscan_launch_config(p->mcols[s], p->mdesc[s], false, c->num_cus);
            if (p->mcfg[s].variant < 0) throw Error(VDL_ERR_UNSUPPORTED, "no multi-aggregate scan kernel variant for this shape");
            std::string jname;
            const bool spec = p->use_jit && !sp.never && specialise_scan(c, p, s, false, &jname);
            if (eligible && !spec) { p->kscan[s] = 1; }       // it did not build: the tuned single-aggregate kernel after all
            p->mparts[s] = dev_alloc(c, sizeof(int64_t) * (size_t)max_scan_grid(c, p, p->mcfg[s].grid) * (size_t)(p->mdesc[s].nagg + 1));
            p->mdesc[s].block_partials = (int64_t *)p->mparts[s]->p;
            if (!p->mdev[s]) p->mdev[s] = dev_alloc(c, sizeof(MScanDesc));
            if (!p->kscan[s]) kname = (spec ? jname : std::string(mscan_kernel_name(p->mcfg[s]))) + "_grid" + std::to_string(p->mcfg[s].grid);
        }
        p->word_offset[s] = off;
        off += (int64_t)sp.aggs.size() + 1;
        if (!sp.never && n * bpr > p->scan_bytes) { p->scan_bytes = n * bpr; p->scan_rows = n; p->dominant = (int)s; p->dominant_kernel = kname; }
    }
    for (size_t g = 0; g < ng; g++) {
        const GroupScanPlan &gp = F.gscans[g];
        const size_t m = ns + g;
        int64_t bpr = 0;
        const int64_t n = bind_mscan(c, gp, p->mcols[m], p->mdesc[m], &bpr, p->row_offset);
        MScanDesc &d = p->mdesc[m];
        d.nkey = (int)gp.key.size();
        for (int k = 0; k < d.nkey; k++) d.key[k] = gp.key[(size_t)k];
        d.pmin = gp.pmin; d.pcount = gp.pcount;
        p->mcfg[m] = mscan_launch_config(p->mcols[m], d, true, c->num_cus);
        if (p->mcfg[m].variant < 0) throw Error(VDL_ERR_UNSUPPORTED, "no grouped-scan kernel variant for this shape");
        std::string jname;
        const bool spec = p->use_jit && !gp.never && specialise_scan(c, p, m, true, &jname);
        const int64_t words = d.pcount * (d.nagg + 1) + 1;
        p->mparts[m] = dev_alloc(c, sizeof(int64_t) * (size_t)max_scan_grid(c, p, p->mcfg[m].grid) * (size_t)words);
        d.block_partials = (int64_t *)p->mparts[m]->p;
        if (!p->mdev[m]) p->mdev[m] = dev_alloc(c, sizeof(MScanDesc));
        p->gword_offset[g] = off;
        off += words;
        if (!gp.never && n * bpr > p->scan_bytes) {
            p->scan_bytes = n * bpr; p->scan_rows = n; p->dominant = (int)m;
            p->dominant_kernel = (spec ? jname : std::string(mscan_kernel_name(p->mcfg[m]))) + "_grid" + std::to_string(p->mcfg[m].grid) + "_rep" + std::to_string(d.replicas);
        }
    }
    p->bound = true;
}

void patch_prelude(const vdl_plan *p, const std::vector<ScanColumn> &sc, MScanCols &cols, MScanDesc &d);
// The columns of a scan with derived columns (fused front, dimension scans) as kernel arguments; the table's row count.
// Tables of the prelude are patched in later (patch_prelude); `wanted` collects which ones.
static int64_t bind_vcols(vdl_ctx *c, const std::string &table, const std::vector<ScanColumn> &sc, MScanCols &cols, MScanDesc &d, std::vector<char> &wanted) {
    cols.ncol = (int)sc.size();
    int64_t n = -1;
    for (int k = 0; k < cols.ncol; k++) {
        const ScanColumn &s = sc[(size_t)k];
        cols.kind[k] = s.kind;
        cols.lo[k] = s.lo; cols.hi[k] = s.hi;
        cols.filtered[k] = (s.lo != INT64_MIN || s.hi != INT64_MAX) ? 1 : 0;
        d.flo[k] = s.lo; d.fhi[k] = s.hi;
        d.dkind[k] = s.kind; d.dsrc[k] = s.idx; d.dsrc2[k] = s.idx2;
        if (s.kind == VC_DIRECT) {
            const Column &col = find_col(c, s.name);
            if (n >= 0 && col.n != n) throw Error(VDL_ERR_SHAPE, "columns of table '" + table + "' have different lengths in the catalog");
            n = col.n;
            cols.ptr[k] = col.dev; cols.width[k] = col.width;
        } else if (s.kind == VC_GATHER || s.kind == VC_INRANGE) {
            const Column &col = find_col(c, s.name);
            cols.ptr[k] = col.dev; cols.width[k] = col.width;
            d.dn[k] = col.n;
        } else {
            cols.ptr[k] = nullptr; cols.width[k] = 8;
            if (s.prelude >= 0) wanted[(size_t)s.prelude] = 1;
        }
    }
    bind_forms(sc, d);
    cols.n = n;
    return n;
}

// The projection scan's passes specialised for this plan (vdl_plan_set_jit): built at the first run after a catalog change,
// kept by role ("select", "take", "dim<k>"); nullptr = the precompiled kernel (not asked for, or it did not build: the note says).
static hipFunction_t front_kernel(vdl_ctx *c, vdl_plan *p, const std::string &role, jit::Kind kind, const MScanCols &cols, const MScanDesc &d) {
    if (!p->use_jit) return nullptr;
    vdl_plan::FrontKernel &fk = p->front_jit[role];
    if (fk.version == c->binding_version()) return fk.k ? fk.k->fn : nullptr;
    {   // a rebuild after a catalog change: this role's old line leaves the note
        const size_t at = p->jit_note.find(role + ": ");
        if (at != std::string::npos) { const size_t end = p->jit_note.find("; ", at); p->jit_note.erase(at, end == std::string::npos ? std::string::npos : end + 2 - at); }
    }
    fk.version = c->binding_version();
    fk.k = nullptr;
    jit::Shape sh;
    sh.nc = cols.ncol; sh.u = 4; sh.vec = kind == jit::SELECT ? project_select_vec(cols) : false; sh.der = true;
    std::vector<char> code;
    std::string why;
    if (jit::compile(jit::scan_source(kind, mscan_args(cols), d, sh), c->arch, code, why)) fk.k = jit::load(code, why, jit::entry_name(kind));
    if (fk.k) p->jit_note += role + ": " + jit::entry_name(kind) + "<" + std::to_string(sh.nc) + ">, " + std::to_string(code.size()) + " B of code; ";
    else p->jit_note += role + ": not specialised (" + why.substr(0, 400) + "); ";
    return fk.k ? fk.k->fn : nullptr;
}

// the one-pass front specialised for this plan: two descriptors in one kernel (jit::front_source)
static hipFunction_t front_kernel_one_pass(vdl_ctx *c, vdl_plan *p, const MScanCols &scols, const MScanDesc &sd, const MScanCols &tcols, const MScanDesc &td) {
    if (!p->use_jit) return nullptr;
    const std::string role = "front";
    vdl_plan::FrontKernel &fk = p->front_jit[role];
    if (fk.version == c->binding_version()) return fk.k ? fk.k->fn : nullptr;
    {
        const size_t at = p->jit_note.find(role + ": ");
        if (at != std::string::npos) { const size_t end = p->jit_note.find("; ", at); p->jit_note.erase(at, end == std::string::npos ? std::string::npos : end + 2 - at); }
    }
    fk.version = c->binding_version();
    fk.k = nullptr;
    jit::Shape sh;
    sh.nc = scols.ncol; sh.u = 4; sh.vec = project_select_vec(scols); sh.der = true;
    std::vector<char> code;
    std::string why;
    if (jit::compile(jit::front_source(mscan_args(scols), sd, mscan_args(tcols), td, sh, tcols.ncol), c->arch, code, why)) fk.k = jit::load(code, why, jit::entry_name(jit::FRONT));
    if (fk.k) p->jit_note += role + ": " + jit::entry_name(jit::FRONT) + "<" + std::to_string(sh.nc) + "," + std::to_string(tcols.ncol) + ">, " + std::to_string(code.size()) + " B of code; ";
    else p->jit_note += role + ": not specialised (" + why.substr(0, 400) + "); ";
    return fk.k ? fk.k->fn : nullptr;
}

// Dimension-side work of scans with derived columns (FusedPlan::prelude): the per-operator executor runs the statements
// that hold the dimension selections (filters on the dimension table, joins of dimensions with further dimensions) and
// their validity bitmaps become the lookup tables of the fact scan; LIKE patterns are evaluated once per heap offset.
// Part of the query: runs on every execution.  `wanted[k]`: item k is referred to by a scan that is about to run.
void run_prelude_items(vdl_ctx *c, vdl_plan *p, const std::vector<char> &asked) {
    const FusedPlan &F = p->fused;
    p->prelude_buf.assign(F.prelude.size(), nullptr);
    p->prelude_n.assign(F.prelude.size(), 0);
    p->prelude_rows.assign(F.prelude.size(), 0);
    std::vector<char> wanted(asked);
    for (size_t k = F.prelude.size(); k-- > 0;)                 // what a wanted dimension scan looks up itself (earlier items)
        if (wanted[k] && F.prelude[k].scan)
            for (const ScanColumn &sc : F.prelude[k].cols) if (sc.prelude >= 0) wanted[(size_t)sc.prelude] = 1;
    const bool scans = !getenv("VDL_NO_DIM_SCAN");
    for (size_t k = 0; k < F.prelude.size(); k++) {
        const PreludeItem &it = F.prelude[k];
        if (!wanted[k] || it.kind != PreludeItem::LIKE_LUT) continue;
        const Column &heap = find_col(c, it.heap);
        Src offs; offs.kind = SRC_RANGE; offs.from = 0; offs.step = 1;
        Src hs; hs.p = heap.dev; hs.kind = heap.width == 8 ? SRC_I64 : heap.width == 4 ? SRC_I32 : heap.width == 2 ? SRC_I16 : SRC_I8;
        LikePattern pat{};
        pat.len = (int)it.pattern.size();
        memcpy(pat.p, it.pattern.data(), it.pattern.size());
        p->prelude_buf[k] = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(heap.n, 1));
        p->prelude_n[k] = heap.n;
        HIP_CHECK(launch_like(offs, nullptr, heap.n, hs, nullptr, heap.n, pat, (int64_t *)p->prelude_buf[k]->p, c->stream));
    }
    std::vector<int> witnesses;
    for (size_t k = 0; k < F.prelude.size(); k++)
        if (wanted[k] && F.prelude[k].kind == PreludeItem::DIM_BITMAP && !(scans && F.prelude[k].scan)) witnesses.push_back(F.prelude[k].witness);
    if (!witnesses.empty()) {
        GenExec g(c, p);
        g.run_nodes(witnesses, nullptr);
        for (size_t k = 0; k < F.prelude.size(); k++) {
            if (!wanted[k] || F.prelude[k].kind != PreludeItem::DIM_BITMAP || (scans && F.prelude[k].scan)) continue;
            const DVec &v = g.vec[(size_t)F.prelude[k].witness];
            p->prelude_n[k] = v.n;
            if (v.kind == DVec::SPARSE) p->prelude_buf[k] = g.bitmap_of(v.sel);
            else p->prelude_buf[k] = g.densify(v).valid;                  // null: every dimension row holds a value
        }
        HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    // dimension scans, in order: an item only looks up earlier ones
    std::vector<BufP> descs;
    // (host copies of the descriptors stay alive in the plan until its next run: the copies to the device are asynchronous)
    std::vector<std::shared_ptr<MScanDesc>> &host_descs = p->host_descs;
    host_descs.clear();
    for (size_t k = 0; k < F.prelude.size(); k++) {
        const PreludeItem &it = F.prelude[k];
        const bool semi = it.kind == PreludeItem::SEMI_BITMAP;
        if (!wanted[k] || !it.scan || !(semi || (scans && it.kind == PreludeItem::DIM_BITMAP))) continue;
        MScanCols cols;
        host_descs.push_back(std::make_shared<MScanDesc>());
        MScanDesc *d = host_descs.back().get();
        std::vector<char> unused(F.prelude.size(), 0);
        const int64_t n = bind_vcols(c, it.table, it.cols, cols, *d, unused);
        patch_prelude(p, it.cols, cols, *d);
        if (semi) {
            // the set of rows of another table that a selected row of this one points at: one scan, atomic ORs
            const int64_t nbits = find_col(c, it.bits_of).n;
            if (it.modulus > 0 && nbits > it.modulus)
                throw NeedGeneralPath("the semi-join's positions are taken mod " + std::to_string(it.modulus) + " but the table they index has " + std::to_string(nbits) + " rows");
            const size_t words = (size_t)std::max<int64_t>((nbits + 63) >> 6, 1);
            p->prelude_buf[k] = dev_alloc(c, sizeof(uint64_t) * words);
            p->prelude_n[k] = nbits;
            HIP_CHECK(hipMemsetAsync(p->prelude_buf[k]->p, 0, sizeof(uint64_t) * words, c->stream));
            p->prelude_rows[k] = std::max<int64_t>(n, 0);
            if (it.never || n <= 0) continue;
            d->bitmap_only = 2;
            d->pmin = it.modulus;
            d->nout = 1; d->out_col[0] = it.index_col;
            // the Scatter this set stands for writes into a vector as long as ITS table (the fold operand, Vlite.hs:1212-1222):
            // positions at or beyond that length are dropped like any out-of-range Scatter position, also when the indexed
            // table is the longer one
            // (sharded: n is this rank's share of the table; the merge clips at the global length: semi_unclamped)
            d->dn[it.index_col] = p->semi_unclamped ? nbits : std::min(nbits, n);
            p->prelude_rows[k] = n;
            d->out_ptr[0] = (int64_t *)p->prelude_buf[k]->p;
            HIP_CHECK(launch_project_select(cols, desc_on_device(c, p, "semi" + std::to_string(k), *d), c->num_cus, c->stream,
                                            front_kernel(c, p, "semi" + std::to_string(k), jit::SELECT, cols, *d)));
            continue;
        }
        const size_t words = (size_t)std::max<int64_t>((n + 63) >> 6, 1);
        p->prelude_buf[k] = dev_alloc(c, sizeof(uint64_t) * words);
        p->prelude_n[k] = n;
        if (it.never || n <= 0) { HIP_CHECK(hipMemsetAsync(p->prelude_buf[k]->p, 0, sizeof(uint64_t) * words, c->stream)); continue; }
        d->out_ptr[0] = (int64_t *)p->prelude_buf[k]->p;           // bitmap only: no positions, no counts
        d->bitmap_only = 1;
        HIP_CHECK(launch_project_select(cols, desc_on_device(c, p, "dim" + std::to_string(k), *d), c->num_cus, c->stream,
                                        front_kernel(c, p, "dim" + std::to_string(k), jit::SELECT, cols, *d)));
    }
}
// hand the tables to a scan that looks them up
void patch_prelude(const vdl_plan *p, const std::vector<ScanColumn> &sc, MScanCols &cols, MScanDesc &d) {
    for (size_t k = 0; k < sc.size(); k++) {
        if (sc[k].kind != VC_BITS && sc[k].kind != VC_LUT) continue;
        const BufP &b = p->prelude_buf[(size_t)sc[k].prelude];
        cols.ptr[k] = b ? b->p : nullptr;
        d.dn[k] = p->prelude_n[(size_t)sc[k].prelude];
    }
}
void run_prelude(vdl_ctx *c, vdl_plan *p) {
    const FusedPlan &F = p->fused;
    if (F.prelude.empty()) return;
    std::vector<char> wanted(F.prelude.size(), 0);
    const size_t ns = F.scans.size();
    for (size_t s = 0; s < ns + F.gscans.size(); s++)
        for (const ScanColumn &sc : (s < ns ? F.scans[s].cols : F.gscans[s - ns].cols))
            if (sc.prelude >= 0) wanted[(size_t)sc.prelude] = 1;
    run_prelude_items(c, p, wanted);
    if (p->after_prelude) p->after_prelude(c, p);
    for (size_t s = 0; s < ns + F.gscans.size(); s++) patch_prelude(p, s < ns ? F.scans[s].cols : F.gscans[s - ns].cols, p->mcols[s], p->mdesc[s]);
}

void run_fused_local(vdl_ctx *c, vdl_plan *p, int64_t *dev_words, bool single_rank) {
    const char *tune_a = getenv("VDL_SCAN_TUNE"), *tune_b = getenv("VDL_GROUP_TUNE");
    if (!p->bound || p->bound_version != c->binding_version() || tune_a || tune_b) {   // tuning sweeps re-bind every run
        bind_fused(c, p);
        p->bound_version = c->binding_version();
    }
    run_prelude(c, p);
    if (p->use_jit && p->jit_tune && !p->jit_tuned) {
        p->jit_tuned = true;
        for (size_t s = 0; s < p->mcols.size(); s++) {            // (the lookup tables of this run are in place)
            const size_t ns0 = p->fused.scans.size();
            if (p->mjit[s]) patch_prelude(p, s < ns0 ? p->fused.scans[s].cols : p->fused.gscans[s - ns0].cols, p->mcols[s], p->mdesc[s]);
        }
        tune_specialised(c, p, dev_words);
    }
    const int ei = (int)(p->run_seq++ % (unsigned)vdl_plan::kEvRing);
    if (p->profiling && !p->ev0[ei]) { HIP_CHECK(hipEventCreate(&p->ev0[ei])); HIP_CHECK(hipEventCreate(&p->ev1[ei])); }
    p->ev_pending[ei] = false;
    p->ev_bound[ei] = false;
    p->ev_seq[ei] = p->run_seq;
    p->last_ev = ei;
    p->ev_buf[ei] = dev_words;
    const size_t ns = p->fused.scans.size(), ng = p->fused.gscans.size();
    for (size_t s = 0; s < ns + ng; s++) {
        const bool grouped = s >= ns;
        const bool never = grouped ? p->fused.gscans[s - ns].never : p->fused.scans[s].never;
        int64_t *out = dev_words + (grouped ? p->gword_offset[s - ns] : p->word_offset[s]);
        const bool kscan = !grouped && p->kscan[s];
        const int64_t n = kscan ? p->sargs[s].n : p->mcols[s].n;
        const bool timed = p->profiling && (int)s == p->dominant && !never && n > 0;
        if (kscan) {
            const ScanArgs &a = p->sargs[s];
            int nblocks = 0;
            if (!never && n > 0) {
                if (timed) HIP_CHECK(hipEventRecord(p->ev0[ei], c->stream));
                HIP_CHECK(launch_scan(a, p->scfg[s], c->stream));
                if (timed) { HIP_CHECK(hipEventRecord(p->ev1[ei], c->stream)); p->ev_pending[ei] = true; }
                nblocks = p->scfg[s].grid;
            }
            HIP_CHECK(launch_scan_finish(a.block_partials, nblocks, a.nagg, nullptr, a, out, c->stream));
        } else {
            const MScanDesc *on_dev = desc_on_device(c, p, "scan" + std::to_string(s), p->mdesc[s]);
            // events bracket the scan together with its tiny finish kernel(s)
            if (timed) HIP_CHECK(hipEventRecord(p->ev0[ei], c->stream));
            HIP_CHECK(launch_mscan(p->mcols[s], p->mdesc[s], on_dev, p->mcfg[s], grouped, never, out,
                                   grouped && single_rank, c->stream, p->mjit[s] ? p->mjit[s]->fn : nullptr));
            if (timed) { HIP_CHECK(hipEventRecord(p->ev1[ei], c->stream)); p->ev_pending[ei] = true; }
        }
    }
}

// Finalisation is split so that callers can pipeline queries: `begin` enqueues the copy of the
// (merged) partial words into a pinned host slot and records an event; `end` waits for that event
// only (not for younger work on the stream) and builds the outputs.
void finalize_begin(vdl_ctx *c, vdl_plan *p, const int64_t *dev_words, int slot) {
    if (!p->bound) throw Error(VDL_ERR_ARG, "vdl_finalize called before vdl_run_local");
    if (slot < 0 || slot > 1) throw Error(VDL_ERR_ARG, "finalisation slot must be 0 or 1");
    if (p->host_cap < p->n_words) {
        for (int k = 0; k < 2; k++) {
            if (p->host_words[k]) HIP_CHECK(hipHostFree(p->host_words[k]));
            HIP_CHECK(hipHostMalloc((void **)&p->host_words[k], sizeof(int64_t) * (size_t)std::max<int64_t>(p->n_words, 1), hipHostMallocDefault));
        }
        p->host_cap = p->n_words;
    }
    if (!p->slot_ev[slot]) HIP_CHECK(hipEventCreateWithFlags(&p->slot_ev[slot], hipEventDisableTiming));
    if (p->n_words) HIP_CHECK(hipMemcpyAsync(p->host_words[slot], dev_words, sizeof(int64_t) * (size_t)p->n_words, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipEventRecord(p->slot_ev[slot], c->stream));
    p->slot_pending[slot] = true;
    int ei = -1;
    for (int k = 0; k < vdl_plan::kEvRing; k++)      // the oldest run into this buffer whose timing nobody has claimed
        if (p->ev_pending[k] && !p->ev_bound[k] && p->ev_buf[k] == (const void *)dev_words && (ei < 0 || p->ev_seq[k] < p->ev_seq[ei])) ei = k;
    if (ei >= 0) p->ev_bound[ei] = true;
    p->slot_ev_idx[slot] = ei;
}

void finalize_end(vdl_ctx *c, vdl_plan *p, int slot) {
    if (slot < 0 || slot > 1 || !p->slot_pending[slot]) throw Error(VDL_ERR_ARG, "no finalisation pending in this slot");
    HIP_CHECK(hipEventSynchronize(p->slot_ev[slot]));
    p->slot_pending[slot] = false;
    const int64_t *wp = p->host_words[slot];
    std::vector<int64_t> w(wp, wp + p->n_words);
    (void)c;
    p->timings.clear();
    if (p->slot_ev_idx[slot] >= 0) {
        const int ei = p->slot_ev_idx[slot];
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, p->ev0[ei], p->ev1[ei]));
        p->scan_usec = (double)ms * 1e3;
        p->timings.push_back({"timeInMicrosecondsForFusedScan_" + p->dominant_kernel, p->scan_usec});
        p->ev_pending[ei] = false;
        p->slot_ev_idx[slot] = -1;
    }
    const size_t ns = p->fused.scans.size();
    for (size_t g = 0; g < p->fused.gscans.size(); g++) {
        const MScanDesc &a = p->mdesc[ns + g];
        const int64_t oob = w[(size_t)(p->gword_offset[g] + a.pcount * (a.nagg + 1))];
        if (oob > 0)
            throw NeedGeneralPath(std::to_string(oob) + " row(s) carry a group key outside the Partition pivots [" + std::to_string(a.pmin) + "," +
                                  std::to_string(a.pmin + a.pcount - 1) + "]");
    }
    p->outs.clear();
    for (const FusedOutput &fo : p->fused.outputs) {
        Output o;
        o.node = fo.node;
        o.name = p->prog.at(fo.node).field;
        o.tmp = "tmp" + std::to_string(fo.node);
        if (fo.gscan < 0) {
            const int64_t *sw = w.data() + p->word_offset[(size_t)fo.scan];
            if (sw[0] > 0) o.vals.push_back(eval_scalar(*fo.value, sw + 1));   // no selected row -> the fold slot is EPS
        } else {
            // one value per non-empty bucket, ascending = the order of the runs of the sorted key
            const MScanDesc &a = p->mdesc[ns + (size_t)fo.gscan];
            const int W = a.nagg + 1;
            const int64_t *tab = w.data() + p->gword_offset[(size_t)fo.gscan];
            for (int64_t b = 0; b < a.pcount; b++)
                if (tab[b * W] > 0) o.vals.push_back(eval_scalar(*fo.value, tab + b * W + 1));
        }
        p->outs.push_back(std::move(o));
    }
}

}  // namespace

namespace vdl {
namespace eng {

// The fused front of a plan that does not fuse as a whole (ProjPlan, vdl_fuse.h): one scan over the fact table -- two
// passes: count per tile, then write -- produces the statements of `proj.nodes` as SPARSE vectors on one shared selection;
// the per-operator executor starts from them (`over`).  false: the plan has no such front (or it is switched off).
// What is known of the fused front before anything runs: the columns of both passes (the select pass sees only the columns
// that decide a row's survival, renumbered), which columns the take pass needs and which it writes.  Pointers and sizes of
// the prelude's tables are patched in by the caller once they exist (patch_front).
struct FrontBound {
    MScanCols cols, scols;
    std::unique_ptr<MScanDesc> d = std::make_unique<MScanDesc>(), sdesc = std::make_unique<MScanDesc>();
    std::vector<int> renum, distinct;
    std::vector<char> wanted;
    int64_t n = 0;
};
static void bind_front(vdl_ctx *c, vdl_plan *p, FrontBound &b) {
    const ProjPlan &J = p->fused.proj;
    MScanCols &cols = b.cols;
    MScanDesc &d = *b.d;
    b.wanted.assign(p->fused.prelude.size(), 0);
    b.n = bind_vcols(c, J.table, J.cols, cols, d, b.wanted);
    cols.row0 = b.scols.row0 = p->row_offset;
    cols.rowid_global = b.scols.rowid_global = p->front_rowid_global || rowid_counts_from_the_table(J.cols);      // (the sharded front route: every row id is the table's)
    // columns that decide a row's survival: the filtered ones and what they are derived from; a lookup whose range check is
    // done by another deciding column through the same index (the dimension bitmap, an INRANGE) decides nothing itself.
    // Everything else is read for the surviving rows only, in the write pass.
    {
        std::vector<char> decides((size_t)cols.ncol, 0);
        for (int k = 0; k < cols.ncol; k++) decides[(size_t)k] = cols.filtered[k] || J.cols[(size_t)k].kind == VC_INRANGE;
        for (int k = 0; k < cols.ncol; k++) {
            const ScanColumn &sc = J.cols[(size_t)k];
            if (sc.kind != VC_GATHER || decides[(size_t)k]) continue;
            bool checked = false;
            for (int j = 0; j < cols.ncol; j++) {
                const ScanColumn &o = J.cols[(size_t)j];
                checked |= j != k && o.idx == sc.idx && (o.kind == VC_INRANGE || o.kind == VC_BITS || (o.kind == VC_GATHER && decides[(size_t)j] && j < k));
            }
            if (!checked) decides[(size_t)k] = 1;                     // its own range check can drop the row
        }
        for (int k = cols.ncol - 1; k >= 0; k--) {
            if (!decides[(size_t)k]) continue;
            for (int src : J.cols[(size_t)k].sources()) decides[(size_t)src] = 1;
        }
        for (int k = 0; k < cols.ncol; k++) cols.lazy[k] = decides[(size_t)k] ? 0 : 1;
    }
    // the select pass: deciding columns only, renumbered (its register use grows with the column count)
    MScanCols &scols = b.scols;
    MScanDesc *sdesc = b.sdesc.get();
    b.renum.assign((size_t)cols.ncol, -1);
    std::vector<int> &renum = b.renum;
    for (int k = 0; k < cols.ncol; k++) {
        if (cols.lazy[k]) continue;
        const int j = scols.ncol++;
        renum[(size_t)k] = j;
        scols.ptr[j] = cols.ptr[k]; scols.width[j] = cols.width[k]; scols.filtered[j] = cols.filtered[k];
        scols.lo[j] = cols.lo[k]; scols.hi[j] = cols.hi[k]; scols.kind[j] = cols.kind[k];
        sdesc->flo[j] = d.flo[k]; sdesc->fhi[j] = d.fhi[k]; sdesc->dkind[j] = d.dkind[k]; sdesc->dn[j] = d.dn[k]; sdesc->dtests[j] = d.dtests[k];
        if (d.dkind[k] == VC_FORM) {                          // its steps stay where they are in the pool; the tests' columns are
            sdesc->dsrc[j] = d.dsrc[k]; sdesc->dsrc2[j] = d.dsrc2[k];      // renumbered (monotonic: they stay sorted by column)
            for (int f = d.dsrc[k]; f < d.dsrc[k] + d.dsrc2[k]; f++) {
                sdesc->form[f] = d.form[f];
                if (d.form[f].op == FormStep::LEAF) sdesc->form[f].col = renum[(size_t)d.form[f].col];
            }
            continue;
        }
        sdesc->dsrc[j] = d.dsrc[k] >= 0 ? renum[(size_t)d.dsrc[k]] : -1;
        sdesc->dsrc2[j] = d.dsrc2[k] >= 0 ? renum[(size_t)d.dsrc2[k]] : -1;
    }
    scols.n = b.n;
    // the take pass: one packed vector per produced column (statements that are the same column share it), and what those
    // columns are derived from
    // (`distinct` holds output codes: a column, or -2 - e for the row expression e the pass evaluates: ProjPlan::exprs)
    for (int oc : J.node_col) if (oc != -1 && std::find(b.distinct.begin(), b.distinct.end(), oc) == b.distinct.end()) b.distinct.push_back(oc);
    d.nout = (int)b.distinct.size();
    d.take = 0;
    d.nexpr = (int)J.exprs.size();
    d.nkey = 0;
    for (size_t e = 0; e < J.exprs.size(); e++) {
        d.expr_at[e] = d.nkey; d.expr_len[e] = (int)J.exprs[e].size();
        for (const KeyStep &st : J.exprs[e]) {
            if (d.nkey >= kMaxKeySteps) throw Error(VDL_ERR_UNSUPPORTED, "internal: the front's expressions exceed the descriptor's step pool");
            d.key[d.nkey++] = st;
            if (st.kind == KeyStep::LOAD) d.take |= 1u << st.col;
        }
    }
    for (size_t o = 0; o < b.distinct.size(); o++) { d.out_col[o] = b.distinct[o]; if (b.distinct[o] >= 0) d.take |= 1u << b.distinct[o]; }
    for (int k = cols.ncol - 1; k >= 0; k--) {
        if (!((d.take >> k) & 1u)) continue;
        for (int src : J.cols[(size_t)k].sources()) d.take |= 1u << src;
    }
    // table columns both sides of the front read (they decide survival AND the outputs need them: the join index of a fact table whose
    // dimension is filtered) stay in LDS for the survivors (MScanDesc::carry)
    d.carry = 0; sdesc->carry = 0;
    int taken = 0;
    for (int k = 0; k < cols.ncol && taken < kMaxCarry; k++) {
        if (cols.lazy[k] || cols.kind[k] != VC_DIRECT || !((d.take >> k) & 1u) || renum[(size_t)k] < 0) continue;
        d.carry |= 1u << k;
        sdesc->carry |= 1u << renum[(size_t)k];
        taken++;
    }
}
// the prelude's tables of this run, in both passes' arguments
static void patch_front(const vdl_plan *p, FrontBound &b) {
    patch_prelude(p, p->fused.proj.cols, b.cols, *b.d);
    for (int k = 0; k < b.cols.ncol; k++) {
        if (b.renum[(size_t)k] < 0) continue;
        b.scols.ptr[b.renum[(size_t)k]] = b.cols.ptr[k];
        b.sdesc->dn[b.renum[(size_t)k]] = b.d->dn[k];
    }
}

bool run_projection(vdl_ctx *c, vdl_plan *p, std::map<int, DVec> &over) {
    const ProjPlan &J = p->fused.proj;
    if (!J.ok || (p->use_fusion && p->fused.ok) || !p->use_fusion || getenv("VDL_NO_PROJECTION")) return false;
    auto fbp = std::make_shared<FrontBound>();                 // (kept by the plan until its next run: its descriptors are copied asynchronously)
    FrontBound &fb = *fbp;
    bind_front(c, p, fb);
    if (fb.scols.ncol > kMaxSelectCols) return false;          // more deciding columns than the select pass takes: statement by statement
    p->front_keep = fbp;
    MScanCols &cols = fb.cols, &scols = fb.scols;
    MScanDesc &d = *fb.d;
    std::unique_ptr<MScanDesc> &sdesc = fb.sdesc;
    const std::vector<char> &wanted = fb.wanted;
    const std::vector<int> &distinct = fb.distinct;
    const int64_t n = fb.n;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (p->profiling) {
        if (!p->stmt_ev[1]) { HIP_CHECK(hipEventCreate(&p->stmt_ev[0])); HIP_CHECK(hipEventCreate(&p->stmt_ev[1])); }
        e0 = p->stmt_ev[0]; e1 = p->stmt_ev[1];
        HIP_CHECK(hipEventRecord(e0, c->stream));
    }
    try {
        run_prelude_items(c, p, wanted);
    } catch (const NeedGeneralPath &e) {
        // an assumption of a set the front looks up does not hold for this catalog (a semi-join's modulus smaller than its
        // table): no front for this run -- the statements run one by one, which is exact for any data
        p->front_keep.reset();
        p->fallback_note = e.what();
        return false;
    }
    patch_front(p, fb);
    SelP sel = std::make_shared<Sel>();
    sel->n = n;
    int64_t m = 0;
    std::vector<BufP> outs(distinct.size());
    if (n > 0 && !J.never) {
        // ONE kernel (vdl_mscan_body.h: project_front_body): deciding columns, survivors' ranks, where they go (look-back over the tiles'
        // counts) and the output vectors.  How long those are is only known afterwards: the kernel writes nothing beyond the capacity it
        // is given and leaves the survivors' number in pinned memory.  From a plan's second run on the capacity is an eighth more than
        // last time; a first run takes the whole table's length when that is cheap and otherwise counts first (capacity 0: no
        // survivor is fetched or written); a run that came up short is repeated with the exact number.
        sel->bitmap = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>((n + 63) >> 6, 1));
        sdesc->out_ptr[0] = (int64_t *)sel->bitmap->p;
        BufP look = dev_alloc(c, (size_t)project_look_bytes(n)), total = dev_alloc(c, sizeof(int64_t));
        int64_t *back = c->flag_words() ? c->flag_words() + 1 : nullptr;       // the last batch posts the survivors' number there (system-scope store)
        auto room_for = [&](int64_t cap) {
            sel->idx = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(cap, 1));
            d.out_idx = (int64_t *)sel->idx->p;
            for (size_t o = 0; o < distinct.size(); o++) {
                outs[o] = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(cap, 1));
                d.out_ptr[o] = (int64_t *)outs[o]->p;
            }
            d.out_cap = cap;
        };
        auto pass = [&]() -> int64_t {
            if (back) *(volatile int64_t *)back = -1;
            HIP_CHECK(launch_project_front(scols, desc_on_device(c, p, "select", *sdesc), cols, desc_on_device(c, p, "take", d), look->p, (int64_t *)total->p, back,
                                           c->num_cus, c->stream, front_kernel_one_pass(c, p, scols, *sdesc, cols, d)));
            // the host goes on as soon as it knows the number -- while the kernel's other batches still fetch their survivors --: what it
            // queues next runs behind the kernel anyway
            int64_t got = 0;
            if (back) { c->wait_flag(back, -1, c->stream); got = *(volatile int64_t *)back; }
            else c->fetch_to_host(total->p, 1, &got, c->stream);
            return got;
        };
        const int64_t per_row = (int64_t)sizeof(int64_t) * (int64_t)(distinct.size() + 1);
        int64_t cap = p->front_m_seen >= 0 ? std::min<int64_t>(n, p->front_m_seen + p->front_m_seen / 8 + 4096)
                                           : (n * per_row <= ((int64_t)1 << 28) ? n : 0);
        room_for(cap);
        m = pass();
        if (m > cap) { room_for(m); m = pass(); }
        p->front_m_seen = m;
    } else {
        sel->idx = dev_alloc(c, sizeof(int64_t));
        for (size_t o = 0; o < distinct.size(); o++) outs[o] = dev_alloc(c, sizeof(int64_t));
    }
    sel->m = m;
    sel->first_slot = -1;
    for (size_t k = 0; k < J.nodes.size(); k++) {
        DVec v;
        v.kind = DVec::SPARSE; v.n = n; v.sel = sel;
        if (J.node_col[k] == -1) { v.data = sel->idx; v.ids = true; }
        else v.data = outs[(size_t)(std::find(distinct.begin(), distinct.end(), J.node_col[k]) - distinct.begin())];
        over[J.nodes[k]] = v;
    }
    p->front_usec = 0;
    if (e0) {
        HIP_CHECK(hipEventRecord(e1, c->stream));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        p->front_usec = (double)ms * 1e3;
    }
    p->front_note = "timeInMicrosecondsForFusedFront_" + J.table + "_" + std::to_string(J.nodes.size()) + "_statement_vectors_" + std::to_string(m) + "_of_" +
                    std::to_string(n) + "_rows_kept_(dimension_side_included)";
    return true;
}

std::string describe_plan(const vdl_plan *p) {
    std::ostringstream o;
    if (p->use_fusion && p->fused.ok) {
        o << "fused: " << p->fused.scans.size() << " scan(s)\n" << describe_fused(p->fused);
        if (p->use_jit) o << "scan kernels specialised for this plan at first run (hiprtc)" << (p->jit_note.empty() ? "" : ": " + p->jit_note) << "\n";
    } else {
        if (!p->fused.ok) o << describe_fused(p->fused);
        if (p->use_jit && p->fused.proj.ok) o << "projection / dimension scans specialised for this plan at first run (hiprtc)" << (p->jit_note.empty() ? "" : ": " + p->jit_note) << "\n";
        else o << "fusion disabled\n";
        o << "general: " << p->prog.order.size() << " statement(s), one kernel per operator\n";
        for (int id : p->prog.order) {
            const Node &n = p->prog.at(id);
            o << "  op " << n.id << " " << op_name(n.op, n.bin) << "\n";
        }
    }
    return o.str();
}

}  // namespace eng
}  // namespace vdl


// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char *vdl_version(void) { return "vdl-mi355x 0.1 (gfx950)"; }

int vdl_open(vdl_ctx **out, int device) {
    if (!out) return VDL_ERR_ARG;
    *out = nullptr;
    vdl_ctx *c = new vdl_ctx();
    int rc = guard(c, [&] {
        c->device = device;
        if (device >= 0) {
            int count = 0;
            if (hipGetDeviceCount(&count) != hipSuccess || device >= count)
                throw Error(VDL_ERR_DEVICE, "HIP device " + std::to_string(device) + " not available (" + std::to_string(count) + " device(s) visible)");
            HIP_CHECK(hipSetDevice(device));
            hipDeviceProp_t prop;
            HIP_CHECK(hipGetDeviceProperties(&prop, device));
            c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            c->arch = prop.gcnArchName;
            if (c->arch.find(':') != std::string::npos) c->arch.resize(c->arch.find(':'));      // "gfx950:sramecc+:xnack-" -> "gfx950"
            if (c->arch.empty()) c->arch = "gfx950";
            HIP_CHECK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
            c->stream = c->own_stream;
        }
    });
    if (rc != VDL_OK) {
        // keep the context so the caller can read the message
        c->device = -1;
    }
    *out = c;
    return rc;
}

void vdl_ctx::wait_flag(int64_t *word, int64_t until_not, hipStream_t s) {
    volatile int64_t *w = word;
    for (unsigned spins = 0; *w == until_not; spins++) {
        __builtin_ia32_pause();
        if ((spins & 0x3fff) == 0x3fff) {                       // every ~16 K polls: is the stream still going?
            const hipError_t e = hipStreamQuery(s);
            if (e == hipSuccess) {                              // idle: the word was written (or never will be: a launch failed)
                if (*w == until_not) throw Error(VDL_ERR_DEVICE, "a kernel that should have reported to the host did not (the stream is idle)");
                break;
            }
            if (e != hipErrorNotReady) throw Error(VDL_ERR_DEVICE, std::string("hipStreamQuery failed: ") + hipGetErrorString(e));
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
}
void vdl_ctx::wait_here(hipStream_t s) {
    int64_t *flag = flag_words();
    if (!flag) { HIP_CHECK(hipStreamSynchronize(s)); return; }
    const int64_t seq = ++post_seq;
    HIP_CHECK(launch_post_words(nullptr, 0, nullptr, flag, seq, s));
    wait_seq(flag, seq, s);
}
void vdl_ctx::fetch_to_host(const void *dev, size_t k, int64_t *out, hipStream_t s) {
    if (k == 0) return;
    int64_t *pin = pinned((int64_t)k), *flag = flag_words();
    if (!pin || !flag) {
        HIP_CHECK(hipMemcpyAsync(out, dev, sizeof(int64_t) * k, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        return;
    }
    const int64_t seq = ++post_seq;
    HIP_CHECK(launch_post_words((const int64_t *)dev, (int64_t)k, pin, flag, seq, s));
    wait_seq(flag, seq, s);
    std::memcpy(out, pin, sizeof(int64_t) * k);
}

void vdl_close(vdl_ctx *c) {
    if (!c) return;
    if (c->device >= 0) {
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        c->comm.reset();
        c->cols.clear();
        c->sorted_state.reset();
        c->pool->trim();
        c->pool->closed = true;
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
        if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
        if (c->copy_ev) (void)hipEventDestroy(c->copy_ev);
    }
    delete c;
}

const char *vdl_last_error(const vdl_ctx *c) { return c ? c->err.c_str() : "null context"; }

int vdl_set_stream(vdl_ctx *c, void *hip_stream) {
    if (!c) return VDL_ERR_ARG;
    return guard(c, [&] {
        need_device(c);
        // pool buffers and plan-owned scratch are protected by stream order alone: drain the stream being left (and the
        // copy stream) so that nothing queued there can still touch memory a launch on the new stream is handed
        if (c->stream != (hipStream_t)hip_stream) {
            HIP_CHECK(hipStreamSynchronize(c->stream));
            if (c->copy_stream) HIP_CHECK(hipStreamSynchronize(c->copy_stream));
        }
        c->stream = (hipStream_t)hip_stream;      // 0 is a real choice: the legacy default stream
    });
}

int vdl_use_own_stream(vdl_ctx *c) {
    if (!c) return VDL_ERR_ARG;
    return guard(c, [&] {
        need_device(c);
        if (c->stream != c->own_stream) {
            HIP_CHECK(hipStreamSynchronize(c->stream));
            if (c->copy_stream) HIP_CHECK(hipStreamSynchronize(c->copy_stream));
        }
        c->stream = c->own_stream;
    });
}

static void check_width(int w) {
    if (w != 1 && w != 2 && w != 4 && w != 8) throw Error(VDL_ERR_ARG, "elem_bytes must be 1, 2, 4 or 8");
}

int vdl_register_column(vdl_ctx *c, const char *name, const void *dev_ptr, int elem_bytes, int64_t nrows) {
    if (!c || !name) return VDL_ERR_ARG;
    return guard(c, [&] {
        check_width(elem_bytes);
        if (nrows < 0 || (!dev_ptr && nrows > 0)) throw Error(VDL_ERR_ARG, "bad column pointer / length");
        if (((uintptr_t)dev_ptr) % (uintptr_t)elem_bytes) throw Error(VDL_ERR_ARG, "column pointer is not aligned to its element size");
        Column col; col.dev = dev_ptr; col.width = elem_bytes; col.n = nrows;
        c->cols[name] = col;
        c->catalog_version++;
    });
}

int vdl_upload_column(vdl_ctx *c, const char *name, const void *host_ptr, int elem_bytes, int64_t nrows) {
    if (!c || !name) return VDL_ERR_ARG;
    return guard(c, [&] {
        need_device(c);
        check_width(elem_bytes);
        if (nrows < 0 || (!host_ptr && nrows > 0)) throw Error(VDL_ERR_ARG, "bad column pointer / length");
        Column col; col.width = elem_bytes; col.n = nrows;
        col.owned = dev_alloc(c, (size_t)nrows * (size_t)elem_bytes);
        col.dev = col.owned->p;
        if (nrows) HIP_CHECK(hipMemcpyAsync(col.owned->p, host_ptr, (size_t)nrows * (size_t)elem_bytes, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        c->cols[name] = col;
        c->catalog_version++;
    });
}

int vdl_generate_column(vdl_ctx *c, const char *name, int elem_bytes, int64_t row0, int64_t nrows, uint64_t seed,
                        int64_t lo, int64_t hi, int64_t mul, int64_t add) {
    if (!c || !name) return VDL_ERR_ARG;
    return guard(c, [&] {
        need_device(c);
        check_width(elem_bytes);
        if (nrows < 0 || hi < lo) throw Error(VDL_ERR_ARG, "bad generator arguments");
        Column col; col.width = elem_bytes; col.n = nrows;
        col.owned = dev_alloc(c, (size_t)nrows * (size_t)elem_bytes);
        col.dev = col.owned->p;
        HIP_CHECK(launch_gen_column(col.owned->p, elem_bytes, row0, nrows, seed, fnv1a(name), lo, hi, mul, add, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        c->cols[name] = col;
        c->catalog_version++;
    });
}

int vdl_drop_column(vdl_ctx *c, const char *name) {
    if (!c || !name) return VDL_ERR_ARG;
    return guard(c, [&] {
        if (!c->cols.erase(name)) throw Error(VDL_ERR_COLUMN, std::string("no column '") + name + "'");
        c->catalog_version++;
    });
}

int vdl_column_info(const vdl_ctx *c, const char *name, int *elem_bytes, int64_t *nrows, const void **dev_ptr) {
    if (!c || !name) return VDL_ERR_ARG;
    auto it = c->cols.find(name);
    if (it == c->cols.end()) return VDL_ERR_COLUMN;
    if (elem_bytes) *elem_bytes = it->second.width;
    if (nrows) *nrows = it->second.n;
    if (dev_ptr) *dev_ptr = it->second.dev;
    return VDL_OK;
}

int vdl_download_column(vdl_ctx *c, const char *name, void *host_ptr, size_t bytes) {
    if (!c || !name || !host_ptr) return VDL_ERR_ARG;
    return guard(c, [&] {
        need_device(c);
        const Column &col = find_col(c, name);
        if (bytes != (size_t)col.n * (size_t)col.width) throw Error(VDL_ERR_ARG, "host buffer size does not match the column");
        if (bytes) HIP_CHECK(hipMemcpyAsync(host_ptr, col.dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
    });
}

int vdl_parse(vdl_ctx *c, const char *text, size_t len, vdl_plan **out) {
    if (!c || !text || !out) return VDL_ERR_ARG;
    *out = nullptr;
    return guard(c, [&] {
        std::unique_ptr<vdl_plan> p(new vdl_plan());
        p->ctx = c;
        p->device = c->device;
        p->prog = parse_program(text, len);
        rewrite_program(p->prog);
        p->fused = fuse_program(p->prog);
        { const char *j = getenv("VDL_JIT"); p->use_jit = j && *j && *j != '0'; p->jit_tune = p->use_jit && atoi(j) >= 2; }
        p->description = describe_plan(p.get());
        *out = p.release();
    });
}

void vdl_plan_free(vdl_plan *p) {
    if (!p) return;
    if (p->device >= 0) (void)hipSetDevice(p->device);
    delete p;
    (void)hipGetLastError();
}

const char *vdl_plan_describe(const vdl_plan *p) { return p ? p->description.c_str() : ""; }
int vdl_plan_is_fused(const vdl_plan *p) { return p && p->use_fusion && p->fused.ok; }
int vdl_plan_set_fusion(vdl_plan *p, int enabled) {
    if (!p) return VDL_ERR_ARG;
    p->use_fusion = enabled != 0;
    p->description = describe_plan(p);
    return VDL_OK;
}
int vdl_plan_set_jit(vdl_plan *p, int enabled) {
    if (!p) return VDL_ERR_ARG;
    if (p->use_jit != (enabled != 0)) p->bound = false;       // the scans are bound again, with or without their specialised kernels
    p->use_jit = enabled != 0;
    p->jit_tune = enabled >= 2;
    p->jit_tuned = false;
    p->description = describe_plan(p);
    return VDL_OK;
}
const char *vdl_plan_jit_note(const vdl_plan *p) { return p ? p->jit_note.c_str() : ""; }
// Builds (hiprtc; no GPU needed) the specialised kernel of every multi-aggregate scan of the plan against the columns
// registered now, without loading or running anything: the note lists each kernel with its code size, or why it failed.
int vdl_plan_jit_check(vdl_ctx *c, vdl_plan *p) {
    if (!c || !p) return VDL_ERR_ARG;
    return guard(c, [&] {
        p->jit_note.clear();
        if (!p->fused.ok && p->fused.proj.ok) {
            // the fused front of a plan that does not fuse as a whole: both passes of the projection scan, and the dimension
            // scans its prelude holds
            auto build = [&](const std::string &role, jit::Kind kind, const MScanCols &cols, const MScanDesc &d) {
                jit::Shape sh;
                sh.nc = cols.ncol; sh.u = 4; sh.vec = kind == jit::SELECT ? project_select_vec(cols) : false; sh.der = true;
                std::vector<char> code;
                std::string log;
                if (!jit::compile(jit::scan_source(kind, mscan_args(cols), d, sh), c->arch, code, log))
                    throw Error(VDL_ERR_UNSUPPORTED, role + " does not build: " + log.substr(0, 2000));
                p->jit_note += role + ": " + jit::entry_name(kind) + "<" + std::to_string(sh.nc) + ">, " + std::to_string(code.size()) + " B of code; ";
            };
            const FusedPlan &F = p->fused;
            for (size_t k = 0; k < F.prelude.size(); k++) {
                if (F.prelude[k].kind != PreludeItem::DIM_BITMAP || !F.prelude[k].scan) continue;
                MScanCols cols;
                auto d = std::make_unique<MScanDesc>();
                std::vector<char> unused(F.prelude.size(), 0);
                bind_vcols(c, F.prelude[k].table, F.prelude[k].cols, cols, *d, unused);
                d->bitmap_only = 1;
                build("dim" + std::to_string(k), jit::SELECT, cols, *d);
            }
            FrontBound fb;
            bind_front(c, p, fb);
            {
                jit::Shape sh;
                sh.nc = fb.scols.ncol; sh.u = 4; sh.vec = project_select_vec(fb.scols); sh.der = true;
                std::vector<char> code;
                std::string log;
                if (!jit::compile(jit::front_source(mscan_args(fb.scols), *fb.sdesc, mscan_args(fb.cols), *fb.d, sh, fb.cols.ncol), c->arch, code, log))
                    throw Error(VDL_ERR_UNSUPPORTED, "front does not build: " + log.substr(0, 2000));
                p->jit_note += std::string("front: ") + jit::entry_name(jit::FRONT) + "<" + std::to_string(sh.nc) + "," + std::to_string(fb.cols.ncol) + ">, " + std::to_string(code.size()) + " B of code; ";
            }
            return;
        }
        if (!p->fused.ok) throw Error(VDL_ERR_UNSUPPORTED, "the plan has no fused scans: " + p->fused.why_not);
        const FusedPlan &F = p->fused;
        const size_t ns = F.scans.size();
        for (size_t s = 0; s < ns + F.gscans.size(); s++) {
            const bool grouped = s >= ns;
            MScanCols cols;
            auto d = std::make_unique<MScanDesc>();
            int64_t bpr = 0;
            if (grouped) {
                const GroupScanPlan &gp = F.gscans[s - ns];
                bind_mscan(c, gp, cols, *d, &bpr, 0);
                d->nkey = (int)gp.key.size();
                for (int k = 0; k < d->nkey; k++) d->key[k] = gp.key[(size_t)k];
                d->pmin = gp.pmin; d->pcount = gp.pcount;
            } else {
                bind_mscan(c, F.scans[s], cols, *d, &bpr, 0);
            }
            const ScanLaunch cfg = mscan_launch_config(cols, *d, grouped, c->num_cus);
            if (cfg.variant < 0) throw Error(VDL_ERR_UNSUPPORTED, "no scan kernel variant for this shape");
            jit::Shape sh = jit_shape(cols, cfg);
            if (getenv("VDL_JIT_CENSUS")) sh.census = true;              // (tests: the measurement build of a staged scan compiles too)
            std::vector<char> code;
            std::string log;
            MsArgs args = mscan_args(cols);
            if (getenv("VDL_JIT_LATE")) args.stages = staged_columns(c, cols, *d, grouped, &args.lazy);      // the staged form of the same scan
            if (getenv("VDL_JIT_LATE") && atoi(getenv("VDL_JIT_LATE")) == 3 && args.lazy) { args.queued = 1; args.stages = 0; }   // ... or its queue form
            if (!jit::compile(jit::mscan_source(args, *d, sh), c->arch, code, log))
                throw Error(VDL_ERR_UNSUPPORTED, "scan " + std::to_string(s) + " does not build: " + log.substr(0, 2000));
            p->jit_note += "scan " + std::to_string(s) + ": " + jit_name(sh) + (args.queued ? " (queue)" : args.lazy ? " (late)" : "") + ", " + std::to_string(code.size()) + " B of code; ";
        }
    });
}
int vdl_plan_set_device_outputs(vdl_plan *p, int enabled) {
    if (!p) return VDL_ERR_ARG;
    p->device_outputs = enabled != 0;
    return VDL_OK;
}
int vdl_output_device(const vdl_plan *p, int k, const int64_t **dev_vals, size_t *n) {
    if (!p || k < 0 || k >= (int)p->outs.size()) return VDL_ERR_ARG;
    const Output &o = p->outs[(size_t)k];
    if (dev_vals) *dev_vals = o.dev;
    if (n) *n = o.count();
    return VDL_OK;
}
int vdl_plan_set_trace(vdl_plan *p, int enabled) {
    if (!p) return VDL_ERR_ARG;
    p->tracing = enabled != 0;
    if (!p->tracing) p->traced.clear();
    return VDL_OK;
}
int vdl_n_traced(const vdl_plan *p) { return p ? (int)p->traced.size() : 0; }
int vdl_traced(const vdl_plan *p, int k, int *node_id, const char **form, int64_t *n, const int64_t **vals, const uint8_t **ok) {
    if (!p || k < 0 || k >= (int)p->traced.size()) return VDL_ERR_ARG;
    const Traced &t = p->traced[(size_t)k];
    if (node_id) *node_id = t.node;
    if (form) *form = t.form;
    if (n) *n = t.n;
    if (vals) *vals = t.have ? t.vals.data() : nullptr;
    if (ok) *ok = t.have ? t.ok.data() : nullptr;
    return VDL_OK;
}
int vdl_plan_set_profiling(vdl_plan *p, int enabled) {
    if (!p) return VDL_ERR_ARG;
    p->profiling = enabled != 0;
    return VDL_OK;
}

int vdl_run(vdl_ctx *c, vdl_plan *p) {
    if (!c || !p) return VDL_ERR_ARG;
    return guard(c, [&] {
        need_device(c);
        if (p->use_fusion && p->fused.ok) {
            const int64_t nw = plan_words(p, nullptr, nullptr);
            if (!p->words || p->words_cap < nw) { p->words = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(nw, 1)); p->words_cap = nw; }
            try {
                run_fused_local(c, p, (int64_t *)p->words->p, true);
                finalize_begin(c, p, (const int64_t *)p->words->p, 0);
                finalize_end(c, p, 0);
                return;
            } catch (const NeedGeneralPath &e) {
                p->fallback_note = e.what();          // exact for any data: rerun statement by statement
            }
        }
        std::map<int, DVec> over;
        bool front = false;
        if (p->after_front) {
            // a sharded run: whatever happens to the front here, this rank takes part in the collective that follows it
            std::string failure;
            try { front = run_projection(c, p, over); } catch (const Error &e) { failure = e.what(); }
            p->after_front(c, p, over, front, failure);
        } else front = run_projection(c, p, over);
        GenExec g(c, p);
        g.run_nodes(p->prog.outputs, front ? &over : nullptr);
        if (front) p->timings.push_back({p->front_note, p->front_usec});
        if (!p->fallback_note.empty()) { p->timings.push_back({"fusedPlanAbandoned: " + p->fallback_note, 0.0}); p->fallback_note.clear(); }
    });
}

int vdl_n_outputs(const vdl_plan *p) { return p ? (int)p->outs.size() : 0; }
int vdl_output(const vdl_plan *p, int k, const char **name, const char **tmp, const int64_t **vals, size_t *n) {
    if (!p || k < 0 || k >= (int)p->outs.size()) return VDL_ERR_ARG;
    const Output &o = p->outs[(size_t)k];
    if (name) *name = o.name.c_str();
    if (tmp) *tmp = o.tmp.c_str();
    if (vals) *vals = o.ptr();
    if (n) *n = o.count();
    return VDL_OK;
}
int vdl_n_timings(const vdl_plan *p) { return p ? (int)p->timings.size() : 0; }
int vdl_timing(const vdl_plan *p, int k, const char **label, double *usec) {
    if (!p || k < 0 || k >= (int)p->timings.size()) return VDL_ERR_ARG;
    if (label) *label = p->timings[(size_t)k].label.c_str();
    if (usec) *usec = p->timings[(size_t)k].usec;
    return VDL_OK;
}
int vdl_plan_scan_stats(const vdl_plan *p, int64_t *rows, int64_t *algo_bytes, double *usec) {
    if (!p) return VDL_ERR_ARG;
    if (rows) *rows = p->scan_rows;
    if (algo_bytes) *algo_bytes = p->scan_bytes;
    if (usec) *usec = p->scan_usec;
    return VDL_OK;
}

int vdl_plan_scan_traffic(vdl_ctx *c, vdl_plan *p, int64_t *bytes_moved, const char **detail) {
    if (!c || !p) return VDL_ERR_ARG;
    return guard(c, [&] {
        need_device(c);
        if (!(p->use_fusion && p->fused.ok)) throw Error(VDL_ERR_UNSUPPORTED, "vdl_plan_scan_traffic serves fused plans");
        p->traffic_detail.clear();
        const int64_t b = scan_bytes_moved(c, p, p->traffic_detail);
        if (bytes_moved) *bytes_moved = b;
        if (detail) *detail = p->traffic_detail.c_str();
    });
}

int vdl_plan_partial_spec(const vdl_plan *p, int64_t *n_words, const int32_t **reduce_ops) {
    if (!p) return VDL_ERR_ARG;
    vdl_plan *q = const_cast<vdl_plan *>(p);
    if (!(p->use_fusion && p->fused.ok)) {
        // not fused: the plan can still be sharded by rows if its outputs hang off global folds (vdl_plan_set_sharded_table)
        std::string why;
        if (!general_partial_spec(p, q->reduce_ops, why)) {
            if (p->ctx) p->ctx->err = "sharded execution needs a fused plan (outputs = global or dense-domain grouped folds) or global folds over the "
                                      "row-sharded table: " + why;
            return VDL_ERR_UNSUPPORTED;
        }
        if (n_words) *n_words = (int64_t)q->reduce_ops.size();
        if (reduce_ops) *reduce_ops = q->reduce_ops.data();
        return VDL_OK;
    }
    q->reduce_ops.clear();
    bool shardable = true;
    const int64_t off = plan_words(p, &q->reduce_ops, &shardable);
    if (!shardable) {
        if (p->ctx) p->ctx->err = "this fused plan builds a semi-join set from every row of a table: it has no sharded route";
        return VDL_ERR_UNSUPPORTED;
    }
    if (n_words) *n_words = off;
    if (reduce_ops) *reduce_ops = q->reduce_ops.data();
    return VDL_OK;
}

int vdl_run_local(vdl_ctx *c, vdl_plan *p, void *dev_partials) {
    if (!c || !p || !dev_partials) return VDL_ERR_ARG;
    return guard(c, [&] {
        need_device(c);
        if (!(p->use_fusion && p->fused.ok)) { general_run_local(c, p, (int64_t *)dev_partials); return; }
        bool shardable = true;
        plan_words(p, nullptr, &shardable);
        if (!shardable) throw Error(VDL_ERR_UNSUPPORTED, "this fused plan builds a semi-join set from every row of a table: it has no sharded route");
        run_fused_local(c, p, (int64_t *)dev_partials, false);
    });
}

int vdl_finalize(vdl_ctx *c, vdl_plan *p, const void *dev_partials) {
    if (!c || !p || !dev_partials) return VDL_ERR_ARG;
    return guard(c, [&] {
        need_device(c);
        if (!(p->use_fusion && p->fused.ok)) { general_finalize(c, p, (const int64_t *)dev_partials); return; }
        finalize_begin(c, p, (const int64_t *)dev_partials, 0);
        finalize_end(c, p, 0);
    });
}

int vdl_plan_set_sharded_table(vdl_plan *p, const char *table) {
    if (!p) return VDL_ERR_ARG;
    p->sharded_table = table ? table : "";
    return VDL_OK;
}
int vdl_plan_set_row_offset(vdl_plan *p, int64_t row0) {
    if (!p) return VDL_ERR_ARG;
    p->row_offset = row0;
    p->bound = false;
    return VDL_OK;
}

int vdl_resolve_first(vdl_ctx *c, vdl_plan *p, void *dev_partials) {
    if (!c || !p || !dev_partials) return VDL_ERR_ARG;
    return guard(c, [&] {
        need_device(c);
        if (!(p->use_fusion && p->fused.ok) || !p->bound) throw Error(VDL_ERR_ARG, "vdl_resolve_first needs a fused plan after vdl_run_local");
        const size_t ns = p->fused.scans.size();
        for (size_t g = 0; g < p->fused.gscans.size(); g++) {
            bool any = false;
            for (const ScanAgg &ag : p->fused.gscans[g].aggs) any |= ag.kind == AGG_FIRST;
            if (!any) continue;
            HIP_CHECK(launch_mscan_resolve_first(p->mcols[ns + g], p->mdesc[ns + g], desc_on_device(c, p, "scan" + std::to_string(ns + g), p->mdesc[ns + g]),
                                                 (int64_t *)dev_partials + p->gword_offset[g], c->stream));
        }
    });
}

int vdl_finalize_begin(vdl_ctx *c, vdl_plan *p, const void *dev_partials, int slot) {
    if (!c || !p || !dev_partials) return VDL_ERR_ARG;
    return guard(c, [&] {
        need_device(c);
        if (!(p->use_fusion && p->fused.ok)) { general_finalize(c, p, (const int64_t *)dev_partials); return; }   // nothing left for _end
        finalize_begin(c, p, (const int64_t *)dev_partials, slot);
    });
}

int vdl_finalize_end(vdl_ctx *c, vdl_plan *p, int slot) {
    if (!c || !p) return VDL_ERR_ARG;
    return guard(c, [&] {
        need_device(c);
        if (!(p->use_fusion && p->fused.ok)) return;
        finalize_end(c, p, slot);
    });
}

}  // extern "C"

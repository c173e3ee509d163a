// vdl_exchange.cpp -- sharded Partition: which vectors travel (analysis) and the three C-ABI calls around the
// caller's all-to-all (vdl_exchange_begin / _pack / _finish); see vdl_partition.hip "Row exchange".
#include "vdl_genexec.h"
#include "vdl_exchange_analysis.h"

static_assert(vdl::kMaxExSourcesAnalysed == vdl::kMaxExSources, "the analysis and the kernels agree on how many vectors one exchange carries");

namespace {

using namespace vdl::exan;       // the analyses: vdl_exchange_analysis.h

// the cut the exchange calls work at: the plan's own, or -- inside the chain route's second stage -- the next Partition with the sets given
ExchangeSpec spec_for(const vdl_plan *p, const std::string &table, bool allow_folds) {
    if (p->chain.stage == 2 && p->chain.plan) return analyse_exchange(with_sets_given(p->prog, p->chain.plan->sets), table, false, true);
    return analyse_exchange(p->prog, table, allow_folds);
}

}  // namespace

namespace vdl {
namespace eng {

size_t exchange_fold_count(const vdl_plan *p, const std::string &table) {
    ExchangeSpec x = spec_for(p, table, true);
    return x.ok ? x.folds.size() : 0;
}
int exchange_fold_kind(const vdl_plan *p, size_t k) { return k < p->ex.folds.size() ? fold_reduce_kind(p->prog.at(p->ex.folds[k]).op) : 0; }

bool general_partial_spec(const vdl_plan *p, std::vector<int32_t> &ops, std::string &why) {
    FoldCut x = analyse_folds(p->prog, p->sharded_table);
    if (!x.ok) { why = x.why; return false; }
    ops.clear();
    for (int id : x.folds) {
        const int k = fold_reduce_kind(p->prog.at(id).op);
        ops.push_back(k == 1 ? VDL_REDUCE_MIN : k == 2 ? VDL_REDUCE_MAX : VDL_REDUCE_SUM);
        ops.push_back(VDL_REDUCE_MIN);          // first control slot (global row id)
        ops.push_back(VDL_REDUCE_SUM);          // number of data folded
    }
    return true;
}

void general_run_local(vdl_ctx *c, vdl_plan *p, int64_t *dev_words) {
    FoldCut x = analyse_folds(p->prog, p->sharded_table);
    if (!x.ok) throw Error(VDL_ERR_UNSUPPORTED, "no sharded route for this plan: " + x.why);
    GenExec g(c, p);
    g.run_nodes(x.folds, nullptr);
    p->cut_folds = x.folds;
    p->cut_n.clear();
    for (size_t k = 0; k < x.folds.size(); k++) {
        const DVec &v = g.vec[(size_t)x.folds[k]];
        if (v.kind != DVec::ONEHOT) throw Error(VDL_ERR_UNSUPPORTED, "global fold " + std::to_string(x.folds[k]) + " did not yield a scalar record");
        p->cut_n.push_back(v.n);
        HIP_CHECK(launch_fold_words((const int64_t *)v.data->p, fold_reduce_kind(p->prog.at(x.folds[k]).op), p->row_offset,
                                    dev_words + 3 * (int64_t)k, c->stream));
    }
    HIP_CHECK(hipStreamSynchronize(c->stream));            // the records die with the executor
}

void general_finalize(vdl_ctx *c, vdl_plan *p, const int64_t *dev_words) {
    if (p->cut_folds.empty()) throw Error(VDL_ERR_ARG, "vdl_finalize before vdl_run_local");
    std::map<int, DVec> over;
    for (size_t k = 0; k < p->cut_folds.size(); k++) {
        DVec v;
        v.kind = DVec::ONEHOT; v.n = p->cut_n[k];
        v.data = dev_alloc(c, 3 * sizeof(int64_t));
        HIP_CHECK(launch_fold_record(dev_words + 3 * (int64_t)k, (int64_t *)v.data->p, c->stream));
        over[p->cut_folds[k]] = v;
    }
    GenExec g(c, p);
    g.run_nodes(p->prog.outputs, &over);
}

}  // namespace eng
}  // namespace vdl

extern "C" {

/* ---- sharded Partition: local phase -> row exchange (caller: RCCL all-to-all) -> local tail ---- */

int vdl_exchange_spec(const vdl_plan *p, const char *sharded_table, int *n_columns) {
    if (!p) return VDL_ERR_ARG;
    ExchangeSpec x = spec_for(p, sharded_table ? sharded_table : "", p->ex_allow_folds);
    if (!x.ok) {
        if (p->ctx) p->ctx->err = "no sharded-Partition structure: " + x.why;
        return VDL_ERR_UNSUPPORTED;
    }
    if (n_columns) *n_columns = (int)x.sources.size() + 1;      // key, scattered vectors, validity mask
    const_cast<vdl_plan *>(p)->sharded_table = sharded_table ? sharded_table : "";     // vdl_exchange_begin analyses for the same placement
    return VDL_OK;
}

}  // extern "C"

namespace vdl {
namespace eng {

std::shared_ptr<ChainPlan> chain_plan(vdl_plan *p, std::string &why) {
    vdl_plan::ChainRun &cr = p->chain;
    if (cr.plan_table != p->sharded_table || (!cr.plan && cr.why.empty())) {
        ChainSpec ch = analyse_chain(p->prog, p->sharded_table);
        cr.plan_table = p->sharded_table;
        cr.plan = ch.ok ? std::make_shared<ChainPlan>(std::move(ch.plan)) : nullptr;
        cr.why = ch.ok ? std::string() : (ch.why.empty() ? std::string("no chain structure") : ch.why);
    }
    why = cr.why;
    return cr.plan;
}

// stage 1 of the chain route has run the GROUP BY on this rank's key range (the targets are alive in `g`): the positions every set
// receives from here, packed -- a position counts where the scattered constant AND the position hold a value
static void chain_collect(GenExec &g, vdl_plan *p) {
    const ChainPlan &cp = *p->chain.plan;
    p->chain.lists.clear();
    for (size_t k = 0; k < cp.sets.size(); k++) {
        const DVec a = g.densify(g.vec[(size_t)cp.targets[3 * k]]), pos = g.densify(g.vec[(size_t)cp.targets[3 * k + 2]]);
        if (a.n != pos.n) throw Error(VDL_ERR_SHAPE, "a position set's constant and positions have different lengths (statement " + std::to_string(cp.sets[k]) + ")");
        vdl_plan::ChainRun::SetList l;
        l.len = g.vec[(size_t)cp.targets[3 * k + 1]].n;
        const BufP bits = (a.valid || pos.valid) ? g.and_bitmaps(a.valid, pos.valid, pos.n) : BufP();
        BufP offsets;
        l.m = g.popcount(bits, pos.n, &offsets);
        l.list = g.compact_write(g.src_of(pos), bits, pos.n, offsets, l.m);
        p->chain.lists.push_back(l);
    }
    HIP_CHECK(hipStreamSynchronize(g.s));                  // (the received rows the vectors may point into go with the caller)
}

// the set as every rank holds it: the constant wherever a position of ANY rank's groups points, `len` slots like the unsharded Scatter's
void chain_build_set(vdl_ctx *c, vdl_plan *p, size_t k, const BufP &positions, int64_t m, int64_t len) {
    const ChainPlan &cp = *p->chain.plan;
    DVec v;
    v.kind = DVec::RANGE; v.n = len; v.from = cp.constant[k]; v.step = 0;
    const int64_t words = std::max<int64_t>(GenExec::nwords(len), 1);
    v.valid = dev_alloc(c, sizeof(uint64_t) * (size_t)words);
    HIP_CHECK(launch_fill_words((uint64_t *)v.valid->p, 0, words, c->stream));
    if (m > 0) HIP_CHECK(launch_set_bits((const int64_t *)positions->p, m, (uint64_t *)v.valid->p, c->stream, len));
    p->chain.sets[cp.sets[k]] = v;
}

void chain_run_everywhere(vdl_ctx *c, vdl_plan *p) {
    need_device(c);
    GenExec g(c, p);
    g.run_nodes(p->prog.outputs, &p->chain.sets);
}

// the local phase of a sharded Partition: the statements up to the key and the scattered vectors on this rank's rows (through the fused
// front when the plan has one), the global folds beside the Partition as mergeable words
void exchange_local(vdl_ctx *c, vdl_plan *p, int world) {
    need_device(c);
    ExchangeSpec x = spec_for(p, p->sharded_table, p->ex_allow_folds);
    if (!x.ok) throw Error(VDL_ERR_UNSUPPORTED, "no sharded-Partition structure: " + x.why);
    std::map<int, DVec> front;                          // the fused front of the local phase (ProjPlan), when the plan has one
    const bool second_stage = p->chain.stage == 2;      // (chain route: the merged sets stand for their statements; the plan's front belongs to stage 1)
    const bool has_front = !second_stage && run_projection(c, p, front);
    GenExec g(c, p);
    std::vector<int> targets = x.sources;
    targets.insert(targets.end(), x.folds.begin(), x.folds.end());
    g.run_nodes(targets, second_stage ? &p->chain.sets : has_front ? &front : nullptr);
    vdl_plan::ExState &ex = p->ex;
    ex = vdl_plan::ExState{};
    ex.world = world; ex.nodes = x.sources; ex.pmin = x.pmin; ex.pcount = x.pcount;
    ex.folds = x.folds;
    if (!x.folds.empty()) {
        // this rank's records of the global folds, as mergeable words {value | identity, first global row | none, count}
        BufP fw = dev_alloc(c, sizeof(int64_t) * 3 * x.folds.size());
        for (size_t k = 0; k < x.folds.size(); k++) {
            DVec v = g.vec[(size_t)x.folds[k]];
            if (v.kind == DVec::OHCONST) v = g.densify(v);
            if (v.kind != DVec::ONEHOT) throw Error(VDL_ERR_UNSUPPORTED, "global fold " + std::to_string(x.folds[k]) + " did not yield a scalar record");
            ex.fold_n.push_back(v.n);
            HIP_CHECK(launch_fold_words((const int64_t *)v.data->p, fold_reduce_kind(p->prog.at(x.folds[k]).op), p->row_offset, (int64_t *)fw->p + 3 * (int64_t)k, c->stream));
        }
        ex.fold_words.resize(3 * x.folds.size());
        c->fetch_to_host(fw->p, ex.fold_words.size(), ex.fold_words.data(), c->stream);
        ex.fold_merged = ex.fold_words;                 // (a single rank: its own records are the merged ones)
    }
    // sources that live on one sparse selection travel as their entries (m rows instead of n slots to route and pack)
    bool all_sparse = !x.sources.empty();
    for (int id : x.sources) {
        const DVec &v = g.vec[(size_t)id];
        all_sparse = all_sparse && v.kind == DVec::SPARSE && v.sel == g.vec[(size_t)x.sources[0]].sel;
    }
    for (int id : x.sources) ex.src.push_back(all_sparse ? g.entries(g.vec[(size_t)id]) : g.densify(g.vec[(size_t)id]));
    ex.n = ex.src[0].n;
    for (const DVec &v : ex.src)
        if (v.n != ex.n) throw Error(VDL_ERR_SHAPE, "vectors scattered by one Partition have different lengths");
}

bool exchange_has_holes(const vdl_plan *p) {
    for (size_t k = 1; k < p->ex.src.size(); k++) if (p->ex.src[k].valid) return true;
    return false;
}

// how this rank's keys spread over kExBins equal slices of the pivots' domain (hist[kExBins] = keys outside the pivots)
void exchange_histogram(vdl_ctx *c, vdl_plan *p, int64_t *hist_host) {
    vdl_plan::ExState &ex = p->ex;
    GenExec g(c, p);
    BufP hist = dev_alloc(c, sizeof(int64_t) * (size_t)(kExBins + 1));
    HIP_CHECK(hipMemsetAsync(hist->p, 0, sizeof(int64_t) * (size_t)(kExBins + 1), c->stream));
    HIP_CHECK(launch_ex_hist(g.src_of(ex.src[0]), g.vp(ex.src[0]), ex.n, ex.pmin, ex.pcount, (int64_t *)hist->p, c->stream));
    c->fetch_to_host(hist->p, (size_t)(kExBins + 1), hist_host, c->stream);
}

// every row's owner -- the declared domain cut evenly (owner_host null: the bare vdl_exchange_* calls), or by the table slice -> rank
// that vdl_run_sharded derives from the ranks' histograms --, the rows' stable order by owner, and the rows per owner
void exchange_route(vdl_ctx *c, vdl_plan *p, const int32_t *owner_host, int64_t *counts_host) {
    vdl_plan::ExState &ex = p->ex;
    const int world = ex.world;
    GenExec g(c, p);
    const DVec &key = ex.src[0];
    BufP counts = dev_alloc(c, sizeof(int64_t) * (size_t)(2 * world + 1));
    HIP_CHECK(hipMemsetAsync(counts->p, 0, sizeof(int64_t) * (size_t)(2 * world + 1), c->stream));
    ex.routecnt = counts;
    ex.owner.reset();
    if (owner_host) {
        ex.owner = dev_alloc(c, sizeof(int32_t) * (size_t)kExBins);
        HIP_CHECK(hipMemcpyAsync(ex.owner->p, owner_host, sizeof(int32_t) * (size_t)kExBins, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));        // (the table is the caller's)
    }
    ExRoute &R = ex.route;
    R = ExRoute{};
    R.key = g.src_of(key); R.vkey = g.vp(key); R.n = ex.n; R.pmin = ex.pmin;
    // (second stage of the chain route: EVERY row with a key travels, to everybody -- the pivots say nothing about who takes part; the
    // Partition that follows treats keys outside them as it does in an unsharded run)
    R.pcount = p->chain.stage == 2 ? 0 : ex.pcount;
    R.world = world; R.shift = ex_route_shift(R.pcount);
    R.owner = ex.owner ? (const int32_t *)ex.owner->p : nullptr;
    // rows per destination and tile, scanned: a row's place in the send buffer is its destination's offset at its tile + its rank among
    // the tile's rows for that destination, which the pack computes again from the key (no destination vector, no positions)
    ex.tileoff = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>((int64_t)world * ex_route_tiles(ex.n), 1));
    HIP_CHECK(launch_ex_route(R, (int64_t *)ex.tileoff->p, (int64_t *)counts->p, c->stream));
    std::vector<int64_t> h((size_t)world + 1);
    c->fetch_to_host(counts->p, (size_t)(world + 1), h.data(), c->stream);
    if (h[(size_t)world] > 0)
        throw Error(VDL_ERR_UNSUPPORTED, std::to_string(h[(size_t)world]) + " row(s) carry a partition key outside the pivots; run unsharded");
    ex.n_send = 0;
    for (int r = 0; r < world; r++) { counts_host[r] = h[(size_t)r]; ex.n_send += h[(size_t)r]; }
    ex.active = true;
}

}  // namespace eng
}  // namespace vdl

extern "C" {

int vdl_exchange_begin(vdl_ctx *c, vdl_plan *p, int world, int64_t *counts_host) {
    if (!c || !p || !counts_host || world < 1 || world > kMaxExWorld) return VDL_ERR_ARG;
    return guard(c, [&] {
        exchange_local(c, p, world);
        exchange_route(c, p, nullptr, counts_host);
    });
}

int vdl_exchange_pack(vdl_ctx *c, vdl_plan *p, void *dev_send) {
    if (!c || !p) return VDL_ERR_ARG;
    return guard(c, [&] {
        need_device(c);
        vdl_plan::ExState &ex = p->ex;
        if (!ex.active) throw Error(VDL_ERR_ARG, "vdl_exchange_pack before vdl_exchange_begin");
        if (ex.n_send == 0) return;
        if (!dev_send) throw Error(VDL_ERR_ARG, "vdl_exchange_pack: null send buffer");
        GenExec g(c, p);
        int64_t *out = (int64_t *)dev_send;
        // every column and the mask of the vectors' validity in one pass over the rows (kExPackCols columns per launch)
        const int ncols = (int)ex.src.size();
        for (int k0 = 0; k0 < ncols; k0 += kExPackCols) {
            ExCols cols;
            cols.first = k0; cols.ncol = std::min(kExPackCols, ncols - k0);
            for (int k = 0; k < cols.ncol; k++) cols.src[k] = g.src_of(ex.src[(size_t)(k0 + k)]);
            if (k0 + kExPackCols >= ncols && !ex.skip_mask) {
                cols.mask_at = ncols;
                for (int k = 1; k < ncols; k++) cols.valid[cols.nvalid++] = g.vp(ex.src[(size_t)k]);
            }
            HIP_CHECK(launch_ex_pack_all(ex.route, cols, (const int64_t *)ex.tileoff->p, (const int64_t *)ex.routecnt->p, ex.n_send, out, c->stream));
        }
        HIP_CHECK(hipStreamSynchronize(c->stream));      // the buffer goes to the caller's collective, possibly on another stream
    });
}

int vdl_exchange_finish(vdl_ctx *c, vdl_plan *p, const void *dev_recv, int64_t n_recv) {
    if (!c || !p || (!dev_recv && n_recv > 0) || n_recv < 0) return VDL_ERR_ARG;
    return guard(c, [&] {
        need_device(c);
        vdl_plan::ExState &ex = p->ex;
        if (!ex.active) throw Error(VDL_ERR_ARG, "vdl_exchange_finish before vdl_exchange_begin");
        const int64_t *in = (const int64_t *)dev_recv;
        const size_t m = ex.src.size();
        std::map<int, DVec> over;
        // usually every travelling row holds a value in every vector (mask word = all ones): no bitmaps needed then
        bool all_valid = n_recv == 0 || m <= 1 || ex.skip_mask;
        if (!all_valid) {
            BufP scratch = dev_alloc(c, sizeof(int64_t) * 3 * (size_t)fold_scratch_blocks());
            BufP r = dev_alloc(c, 3 * sizeof(int64_t));
            Src mk; mk.p = in + (int64_t)m * n_recv; mk.kind = SRC_I64;
            HIP_CHECK(launch_fold_global(1 /* min */, mk, nullptr, nullptr, n_recv, (int64_t *)scratch->p, (int64_t *)r->p, c->stream));
            int64_t h[3];
            c->fetch_to_host(r->p, 3, h, c->stream);
            all_valid = h[0] == (int64_t)(((uint64_t)1 << (m - 1)) - 1);
        }
        for (size_t k = 0; k < m; k++) {
            DVec v;
            v.kind = DVec::COLUMN; v.n = n_recv; v.ptr = in + (int64_t)k * n_recv; v.width = 8;
            if (k > 0 && !all_valid) {      // source vectors had EPS rows: rebuild their bitmaps from the mask column
                v.valid = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(GenExec::nwords(n_recv), 1));
                HIP_CHECK(launch_ex_unmask(in + (int64_t)m * n_recv, n_recv, (int)k - 1, (uint64_t *)v.valid->p, c->stream));
            }
            over[ex.nodes[k]] = v;
        }
        for (size_t k = 0; k < ex.folds.size(); k++) {             // the merged records of the global folds beside the Partition
            DVec v;
            v.kind = DVec::ONEHOT; v.n = std::max<int64_t>(ex.fold_n[k], 1);
            BufP w = dev_alloc(c, 3 * sizeof(int64_t));
            v.data = dev_alloc(c, 3 * sizeof(int64_t));
            HIP_CHECK(hipMemcpyAsync(w->p, ex.fold_merged.data() + 3 * k, 3 * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
            HIP_CHECK(launch_fold_record((const int64_t *)w->p, (int64_t *)v.data->p, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));            // (`w` and the host words may go)
            over[ex.folds[k]] = v;
        }
        ex.src.clear(); ex.tileoff.reset(); ex.owner.reset(); ex.routecnt.reset();      // phase-A vectors are no longer needed
        ex.active = false;
        if (p->chain.stage == 2) over.insert(p->chain.sets.begin(), p->chain.sets.end());
        GenExec g(c, p);
        g.run_nodes(p->prog.outputs, &over);
        if (p->chain.stage == 1) chain_collect(g, p);
    });
}

}  // extern "C"

// vdl_genexec.h -- the per-operator executor (one kernel per VDL statement, vectors in the cheapest form that
// represents them).  Included by vdl_engine.cpp (vdl_run) and vdl_exchange.cpp (the two phases of a sharded Partition).
#pragma once
#include "vdl_engine_internal.h"

namespace vdl {
namespace eng {

// ------------------------------------------------------------------------------------------------
// general (per-operator) execution
// ------------------------------------------------------------------------------------------------
struct GenExec {
    vdl_ctx *c;
    vdl_plan *p;
    std::vector<DVec> vec;
    std::vector<int> last_use;
    hipStream_t s;

    GenExec(vdl_ctx *ctx, vdl_plan *plan) : c(ctx), p(plan), vec(plan->prog.nodes.size()), last_use(plan->prog.nodes.size(), -1), s(ctx->stream) {}
    ~GenExec() { if (!copies_in_flight.empty() && c->copy_stream) (void)hipStreamSynchronize(c->copy_stream); }     // error exits

    static int64_t nwords(int64_t n) { return (n + 63) >> 6; }
    const uint64_t *vp(const DVec &v) const { return v.valid ? (const uint64_t *)v.valid->p : nullptr; }

    Src src_of(const DVec &v) const {
        Src r;
        switch (v.kind) {
        case DVec::DENSE: r.p = v.data->p; r.kind = SRC_I64; break;
        case DVec::COLUMN:
            r.p = v.ptr;
            r.kind = v.width == 8 ? SRC_I64 : v.width == 4 ? SRC_I32 : v.width == 2 ? SRC_I16 : SRC_I8;
            break;
        case DVec::RANGE: r.kind = SRC_RANGE; r.from = v.from; r.step = v.step; break;
        default: throw Error(VDL_ERR_UNSUPPORTED, "internal: operand form not addressable");
        }
        return r;
    }

    // ---- validity bitmaps: which ones are known to be subsets of which (filters nest) ------------------
    std::map<const void *, std::set<const void *>> supers;      // bitmap -> bitmaps known to contain it
    std::map<const void *, SelP> sel_of_bitmap;                 // bitmaps whose population / slot list is known
    std::vector<BufP> keep_alive;                               // keys above stay valid for the whole run
    bool sparse_on = !getenv("VDL_NO_SPARSE");
    bool trace_forms = getenv("VDL_TRACE_FORMS") && std::strcmp(getenv("VDL_TRACE_FORMS"), "0") != 0;      // one line per statement: the form of its result
    int densified = 0;                                             // SPARSE -> DENSE conversions (each is a scatter over n slots)

    bool subset(const BufP &a, const BufP &b) {                 // a (null = all slots) inside b?
        if (!b || a == b) return true;
        if (!a) return false;
        auto it = supers.find(a->p);
        return it != supers.end() && it->second.count(b->p);
    }
    void note_subset(const BufP &child, const BufP &parent) {
        if (!child || !parent || child == parent) return;
        keep_alive.push_back(child); keep_alive.push_back(parent);
        std::set<const void *> &sup = supers[child->p];
        sup.insert(parent->p);
        auto it = supers.find(parent->p);
        if (it != supers.end()) sup.insert(it->second.begin(), it->second.end());
    }
    BufP and_bitmaps(const BufP &a, const BufP &b, int64_t n) {
        if (subset(a, b)) return a;
        if (subset(b, a)) return b;
        BufP o = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(n), 1));
        HIP_CHECK(launch_and_words((const uint64_t *)a->p, (const uint64_t *)b->p, (uint64_t *)o->p, nwords(n), s));
        note_subset(o, a); note_subset(o, b);
        return o;
    }
    BufP and_valid(const DVec &a, const DVec &b, int64_t n) { return and_bitmaps(a.valid, b.valid, n); }

    // ---- sparse vectors --------------------------------------------------------------------------------
    BufP zero_bitmap(int64_t n) {
        BufP o = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(n), 1));
        HIP_CHECK(launch_fill_words((uint64_t *)o->p, 0, nwords(n), s));
        return o;
    }
    // k (<= 64) int64 words from the device to the host, waited for: through the context's pinned words (a pageable destination is
    // staged by the runtime and keeps the GPU idle two to three times as long)
    void fetch_words(const void *dev, int k, int64_t *out) {
        c->fetch_to_host(dev, (size_t)k, out, s);
    }
    // population of a bitmap over n slots; leaves the per-tile offsets for compact_write in *offsets
    int64_t popcount(const BufP &bits, int64_t n, BufP *offsets) {
        const int64_t nb = (n + compact_tile() - 1) / compact_tile();
        if (nb <= 0) return 0;
        BufP counts = dev_alloc(c, sizeof(int64_t) * (size_t)(nb + 1));
        int64_t total = 0;
        if (int64_t *pin = c->pinned(1)) {                     // the kernel leaves the total in pinned host memory itself: no copy to launch
            HIP_CHECK(launch_compact_offsets(bits ? (const uint64_t *)bits->p : nullptr, n, (int64_t *)counts->p, s, pin));
            c->wait_here(s);
            total = *(volatile int64_t *)pin;
        } else {
            HIP_CHECK(launch_compact_offsets(bits ? (const uint64_t *)bits->p : nullptr, n, (int64_t *)counts->p, s));
            fetch_words((int64_t *)counts->p + nb, 1, &total);
        }
        if (offsets) *offsets = counts;
        return total;
    }
    // the entries of `v` (any addressable form, length n) where `bits` is set, packed (total > 0 of them)
    BufP compact_write(Src v, const BufP &bits, int64_t n, const BufP &offsets, int64_t total) {
        BufP out = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(total, 1));
        if (total > 0)
            HIP_CHECK(launch_compact_write(v, bits ? (const uint64_t *)bits->p : nullptr, n, (const int64_t *)offsets->p, (int64_t *)out->p, s));
        return out;
    }
    static Src iota_src() { Src r; r.kind = SRC_RANGE; r.from = 0; r.step = 1; return r; }
    static Src i64_src(const BufP &b) { Src r; r.p = b ? b->p : nullptr; r.kind = SRC_I64; return r; }
    Src idx_src(const Sel &sel) const { return sel.idx ? i64_src(sel.idx) : iota_src(); }

    // the selection a validity bitmap describes (population counted once per bitmap); worth = sparse enough to compact
    SelP sel_for(const BufP &bits, int64_t n) {
        auto it = sel_of_bitmap.find(bits->p);
        if (it != sel_of_bitmap.end()) return it->second;
        SelP sel = std::make_shared<Sel>();
        sel->n = n; sel->bitmap = bits;
        BufP offsets;
        sel->m = popcount(bits, n, &offsets);
        // a subset of a known selection with the same population IS that selection (two routes to one filter)
        auto sup = supers.find(bits->p);
        if (sup != supers.end())
            for (const void *q : sup->second) {
                auto known = sel_of_bitmap.find(q);
                if (known != sel_of_bitmap.end() && known->second->n == n && known->second->m == sel->m && known->second->bitmap) {
                    keep_alive.push_back(bits);
                    sel_of_bitmap[bits->p] = known->second;
                    note_subset(known->second->bitmap, bits);
                    return known->second;
                }
            }
        const char *den = getenv("VDL_SPARSE_DEN");                                 // compaction threshold m <= n / den (default 8)
        sel->worth = sel->m * (den && atoi(den) > 0 ? atoi(den) : 8) <= n || getenv("VDL_SPARSE_ALWAYS") != nullptr;      // the env switch makes the tests cover every path
        if (sel->worth) sel->idx = compact_write(iota_src(), bits, n, offsets, sel->m);
        keep_alive.push_back(bits);
        sel_of_bitmap[bits->p] = sel;
        return sel;
    }
    const BufP &bitmap_of(const SelP &sel) {                     // derived / prefix selections get their bitmap on first use
        if (!sel->bitmap) {
            sel->bitmap = zero_bitmap(sel->n);
            if (sel->m > 0 && !sel->idx) {
                // a prefix selection: the first m bits, written as words (setting 18 M bits one id at a time took 1.4 ms)
                HIP_CHECK(launch_fill_words((uint64_t *)sel->bitmap->p, ~0ull, sel->m >> 6, s));
                if (sel->m & 63) HIP_CHECK(launch_fill_words((uint64_t *)sel->bitmap->p + (sel->m >> 6), (1ull << (sel->m & 63)) - 1, 1, s));
            } else if (sel->m > 0) {
                HIP_CHECK(launch_set_bits((const int64_t *)sel->idx->p, sel->m, (uint64_t *)sel->bitmap->p, s));
            }
            keep_alive.push_back(sel->bitmap);
            sel_of_bitmap[sel->bitmap->p] = sel;
            if (sel->parent) note_subset(sel->bitmap, bitmap_of(sel->parent));
        }
        return sel->bitmap;
    }
    int64_t first_slot_of(const SelP &sel) {
        if (!sel->idx || sel->m <= 0) return 0;
        if (sel->first_slot < 0) {
            fetch_words(sel->idx->p, 1, &sel->first_slot);
        }
        return sel->first_slot;
    }
    struct GatherLive { BufP pos, fvalid, sub, offsets; SelP sel, child; int64_t live = 0; };      // (every buffer of the key is kept alive by the entry)
    std::map<std::tuple<const void *, const void *, int64_t>, GatherLive> gather_live;
    struct RunHeads { BufP ctl, heads, wordhd, offsets; int64_t count = 0; SelP sel, child; };
    // run heads a Partition's sortedness pass left behind (partition_positions): when the key turns out to be in order, the folds of
    // the GROUP BY run over that very buffer and need neither a head pass nor a count of their own
    struct SortedHeads { BufP key, heads, offsets; int64_t count = 0; };
    std::map<const void *, SortedHeads> sorted_heads;           // key entries buffer -> (kept alive by the entry)
    bool group_batch_on = !getenv("VDL_NO_GROUP_BATCH");        // all folds of one GROUP BY in one launch (launch_group_fold)
    std::vector<char> done;                                     // statements executed ahead of their turn (ensure)
    const std::map<int, DVec> *cur_over = nullptr;
    std::vector<char> needed_now;
    std::map<std::pair<const void *, const void *>, RunHeads> heads_of;   // (control entries buffer, its selection) -> run heads (both kept alive by the entry)
    struct DenseHeads { BufP ctl, ctlv, heads, wordhd; };
    std::map<std::pair<const void *, const void *>, DenseHeads> dense_heads;    // (control data, its validity) of a stored control vector -> run heads
    SelP prefix_selection(int64_t n, int64_t m) {
        for (const SelP &x : prefixes) if (x->n == n && x->m == m) return x;
        SelP x = std::make_shared<Sel>();
        x->n = n; x->m = m;
        prefixes.push_back(x);
        return x;
    }
    std::vector<SelP> prefixes;
    // the selection made of the entries of `ps` flagged in `flags` (a bitmap over its m entries)
    SelP child_selection(const SelP &ps, const BufP &flags, int64_t count, const BufP &offsets) {
        SelP ch = std::make_shared<Sel>();
        ch->n = ps->n; ch->parent = ps; ch->m = count;
        // the selected entries' slot ids and their entry numbers inside the parent: one launch writes both
        ch->idx = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(count, 1));
        ch->ppos = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(count, 1));
        if (count > 0)
            HIP_CHECK(launch_compact_write(idx_src(*ps), flags ? (const uint64_t *)flags->p : nullptr, ps->m, (const int64_t *)offsets->p, (int64_t *)ch->idx->p, s, (int64_t *)ch->ppos->p));
        if (!ps->idx && ps->m == ps->n && flags && !sel_of_bitmap.count(flags->p)) {
            // the parent is ALL n slots: the flags over its entries are the child's bitmap over the slots as they stand
            ch->bitmap = flags;
            keep_alive.push_back(flags);
            sel_of_bitmap[flags->p] = ch;
        }
        return ch;                                              // its n-bit bitmap is built when somebody asks (bitmap_of)
    }
    // SPARSE vectors hold a value in every entry: entries that turned EPS (a gather out of range / from an EPS
    // slot) leave the selection
    DVec sparse_normalised(const SelP &sel, const BufP &data, const BufP &sub) {
        BufP offsets;
        const int64_t live = popcount(sub, sel->m, &offsets);
        if (live == sel->m) return make_sparse(sel, data);
        SelP ch = child_selection(sel, sub, live, offsets);
        return make_sparse(ch, compact_write(i64_src(data), sub, sel->m, offsets, live));
    }
    // m-entry view of a SPARSE vector as an ordinary dense one (for the kernels that take Src + bitmap + length)
    DVec entries(const DVec &v) const {
        DVec o; o.kind = DVec::DENSE; o.n = v.sel->m; o.data = v.data;
        return o;
    }
    DVec make_sparse(const SelP &sel, BufP data) {
        DVec o; o.kind = DVec::SPARSE; o.n = sel->n; o.sel = sel; o.data = std::move(data);
        return o;
    }
    // values of a dense-form vector on a selection (its validity must cover the selection)
    DVec sparse_take(const DVec &src, const SelP &sel) {
        BufP data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(sel->m, 1));
        BufP junk = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(sel->m), 1));
        HIP_CHECK(launch_gather(src_of(src), nullptr, src.n, idx_src(*sel), nullptr, sel->m, (int64_t *)data->p, (uint64_t *)junk->p, s));
        return make_sparse(sel, data);
    }
    // SPARSE -> DENSE + bitmap over the n slots
    DVec sparse_to_dense(const DVec &v) {
        densified++;
        DVec o; o.kind = DVec::DENSE; o.n = v.n;
        o.data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(v.n, 1));
        if (v.sel->bitmap) {                                   // the slots that get written are exactly the selection's bitmap
            o.valid = v.sel->bitmap;
            HIP_CHECK(launch_scatter(i64_src(v.data), nullptr, idx_src(*v.sel), nullptr, v.sel->m, v.n, (int64_t *)o.data->p, nullptr, s));
        } else {
            o.valid = zero_bitmap(v.n);
            HIP_CHECK(launch_scatter(i64_src(v.data), nullptr, idx_src(*v.sel), nullptr, v.sel->m, v.n, (int64_t *)o.data->p, (uint64_t *)o.valid->p, s));
            v.sel->bitmap = o.valid; keep_alive.push_back(o.valid); sel_of_bitmap[o.valid->p] = v.sel;
        }
        return o;
    }

    // ---- fused element-wise trees -------------------------------------------------------------------------
    bool fuse_on = !getenv("VDL_NO_EXPR_FUSION");
    std::vector<int> n_uses;                 // readers of each statement in this run (targets count as one more)
    std::vector<int> value_uses;             // ... of its VALUES: RangeV readers, which only want its shape, left out
    std::vector<char> value_read_by_binary_only;
    std::vector<char> read_by_binary_only;
    static bool leafable(const DVec &v) { return v.kind == DVec::DENSE || v.kind == DVec::COLUMN || v.kind == DVec::RANGE; }
    std::shared_ptr<ExprNode> expr_of(const DVec &v) {
        if (v.kind == DVec::EXPR) return v.ex;
        auto e = std::make_shared<ExprNode>();
        e->leaf = v;
        return e;
    }
    BufP expr_valid(const ExprNode &e, int64_t n) {               // AND of the leaves' validity (null = all slots)
        if (e.bin < 0) return e.leaf.valid;
        BufP a = expr_valid(*e.l, n), b = expr_valid(*e.r, n);
        if (!a) return b;
        if (!b) return a;
        return and_bitmaps(a, b, n);
    }
    void expr_emit(const ExprNode &e, ExprProg &prog) {
        if (e.bin < 0) {
            const Src sv = src_of(e.leaf);
            int at = -1;
            for (int k = 0; k < prog.n_leaf; k++)
                if (prog.leaf[k].p == sv.p && prog.leaf[k].kind == sv.kind && prog.leaf[k].from == sv.from && prog.leaf[k].step == sv.step) at = k;
            if (at < 0) { at = prog.n_leaf++; prog.leaf[at] = sv; }
            prog.code[prog.n_instr++] = (signed char)(-at - 1);
            return;
        }
        // the emitter's sugar recognised back (Vdl.hs:139-152): a >= b arrives as LogicalOr(Greater(a,b), Equals(b,a)) (in either
        // operand order), a != b as Subtract(1, Equals(a,b))
        auto same_leaf = [&](const ExprNode &x, const ExprNode &y) {
            if (x.bin >= 0 || y.bin >= 0) return false;
            const Src p = src_of(x.leaf), q = src_of(y.leaf);
            return p.p == q.p && p.kind == q.kind && p.from == q.from && p.step == q.step && x.leaf.n == y.leaf.n;
        };
        if (e.bin == B_LOR && e.l->bin == B_GT && e.r->bin == B_EQ) {
            const ExprNode &g = *e.l, &q = *e.r;
            if ((same_leaf(*g.l, *q.r) && same_leaf(*g.r, *q.l)) || (same_leaf(*g.l, *q.l) && same_leaf(*g.r, *q.r))) {
                expr_emit(*g.l, prog); expr_emit(*g.r, prog);
                prog.code[prog.n_instr++] = (signed char)X_GE;
                return;
            }
        }
        if (e.bin == B_SUB && e.l->bin < 0 && e.l->leaf.kind == DVec::RANGE && e.l->leaf.step == 0 && e.l->leaf.from == 1 && e.r->bin == B_EQ) {
            expr_emit(*e.r->l, prog); expr_emit(*e.r->r, prog);
            prog.code[prog.n_instr++] = (signed char)X_NE;
            return;
        }
        expr_emit(*e.l, prog);
        expr_emit(*e.r, prog);
        prog.code[prog.n_instr++] = (signed char)e.bin;
    }
    static bool expr_fits(const ExprNode &e) { return e.leaves <= kExprLeaves && e.instrs <= kExprInstrs && e.depth <= kExprDepth; }
    DVec expr_force(const DVec &v) {
        if (!expr_fits(*v.ex)) {            // a predicate tree larger than k_expr takes (kept for k_pred) has to produce values after all
            auto side = [&](const std::shared_ptr<ExprNode> &x) {
                if (x->bin < 0) return x;
                DVec t; t.kind = DVec::EXPR; t.n = v.n; t.ex = x;
                auto leaf = std::make_shared<ExprNode>();
                leaf->leaf = expr_force(t);
                return leaf;
            };
            auto top = std::make_shared<ExprNode>();
            top->bin = v.ex->bin; top->l = side(v.ex->l); top->r = side(v.ex->r);
            top->leaves = 2; top->instrs = 3; top->depth = 2;
            DVec t; t.kind = DVec::EXPR; t.n = v.n; t.ex = top;
            return expr_force(t);
        }
        const ExprNode &e = *v.ex;
        DVec o; o.kind = DVec::DENSE; o.n = v.n;
        o.valid = expr_valid(e, v.n);
        o.data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(v.n, 1));
        ExprProg prog;
        expr_emit(e, prog);
        HIP_CHECK(launch_expr(prog, (int64_t *)o.data->p, v.n, s));
        return o;
    }
    // A tree that is a filter predicate -- LogicalAnd / LogicalOr over comparisons of stored vectors -- as a program on
    // 64-row masks (k_pred): no 0/1 vector is ever written.  false: not of that shape (arithmetic inside, too large).
    bool pred_on = !getenv("VDL_NO_PRED_FUSION");
    std::vector<char> lazy_pred_ok;          // Binary statements read only by the filter idiom FoldSelect(RangeV 0 1 this, this)
    bool pred_cmp(PredProg &P, int op, const DVec &a, const DVec &b) {
        if (P.n_cmp >= kPredCmps || P.n_instr >= kPredInstrs || a.n != b.n) return false;
        P.a[P.n_cmp] = src_of(a); P.b[P.n_cmp] = src_of(b); P.op[P.n_cmp] = (signed char)op;
        P.code[P.n_instr++] = (signed char)P.n_cmp++;
        return true;
    }
    bool pred_emit(const ExprNode &e, PredProg &P, int depth) {
        if (depth >= kPredDepth) return false;
        auto is_leaf = [](const ExprNode &x) { return x.bin < 0; };
        auto same_leaf = [&](const ExprNode &x, const ExprNode &y) {
            if (x.bin >= 0 || y.bin >= 0) return false;
            const Src p = src_of(x.leaf), q = src_of(y.leaf);
            return p.p == q.p && p.kind == q.kind && p.from == q.from && p.step == q.step && x.leaf.n == y.leaf.n;
        };
        if (is_leaf(e)) {                                        // a stored value read as a truth value
            DVec zero; zero.kind = DVec::RANGE; zero.n = e.leaf.n; zero.from = 0; zero.step = 0;
            return pred_cmp(P, P_NE, e.leaf, zero);
        }
        if ((e.bin == B_GT || e.bin == B_EQ) && is_leaf(*e.l) && is_leaf(*e.r))
            return pred_cmp(P, e.bin == B_GT ? P_GT : P_EQ, e.l->leaf, e.r->leaf);
        if (e.bin == B_LOR && e.l->bin == B_GT && e.r->bin == B_EQ) {        // a >= b as the emitter prints it (Vdl.hs:143)
            const ExprNode &g = *e.l, &q = *e.r;
            if ((same_leaf(*g.l, *q.r) && same_leaf(*g.r, *q.l)) || (same_leaf(*g.l, *q.l) && same_leaf(*g.r, *q.r)))
                return pred_cmp(P, P_GE, g.l->leaf, g.r->leaf);
        }
        if (e.bin == B_SUB && is_leaf(*e.l) && e.l->leaf.kind == DVec::RANGE && e.l->leaf.step == 0 && e.l->leaf.from == 1 &&
            e.r->bin == B_EQ && is_leaf(*e.r->l) && is_leaf(*e.r->r))            // a != b = 1 - (a == b) (Vdl.hs:152)
            return pred_cmp(P, P_NE, e.r->l->leaf, e.r->r->leaf);
        if (e.bin == B_LAND || e.bin == B_LOR) {
            if (!pred_emit(*e.l, P, depth) || !pred_emit(*e.r, P, depth + 1)) return false;
            if (P.n_instr >= kPredInstrs) return false;
            P.code[P.n_instr++] = (signed char)(e.bin == B_LAND ? P_AND : P_OR);
            return true;
        }
        return false;
    }

    // Binary over stored vectors / pending trees: extend the tree; run it unless the only reader is another Binary
    bool expr_binary(const Node &n, const DVec &a, const DVec &b, DVec &o) {
        if (!fuse_on || n.bin == B_DIV || n.bin == B_MOD) return false;      // division stays in k_binary (code size, see k_expr)
        const bool ta = a.kind == DVec::EXPR, tb = b.kind == DVec::EXPR;
        if (!(ta || leafable(a)) || !(tb || leafable(b)) || a.n != b.n) return false;
        if (!ta && !tb && a.kind == DVec::RANGE && b.kind == DVec::RANGE && a.step == 0 && b.step == 0) return false;   // constant folding stays
        // (RangeV readers take the shape only -- exec(RangeV) reads the validity of a pending tree's leaves -- so they do not end a chain)
        const bool lazy = (value_uses[(size_t)n.id] == 1 && value_read_by_binary_only[(size_t)n.id]) || (pred_on && lazy_pred_ok[(size_t)n.id]);
        if (!lazy && !ta && !tb) return false;                   // a lone operator: the plain kernel
        DVec x = a, y = b;
        for (;;) {
            auto el = expr_of(x), er = expr_of(y);
            auto t = std::make_shared<ExprNode>();
            t->bin = n.bin; t->l = el; t->r = er;
            t->leaves = el->leaves + er->leaves;
            t->instrs = el->instrs + er->instrs + 1;
            t->depth = std::max(el->depth, er->depth + 1);
            bool fits = expr_fits(*t);
            if (!fits && pred_on && (n.bin == B_LAND || n.bin == B_LOR)) {      // predicates have limits of their own (k_pred)
                PredProg probe;
                fits = pred_emit(*t, probe, 0);
            }
            if (fits) {
                o = DVec{};
                o.kind = DVec::EXPR; o.n = a.n; o.ex = t;
                if (!lazy) o = expr_force(o);
                return true;
            }
            // too big for one kernel: run the larger side now and keep it as a leaf
            if (x.kind == DVec::EXPR && (y.kind != DVec::EXPR || el->instrs >= er->instrs)) x = expr_force(x);
            else if (y.kind == DVec::EXPR) y = expr_force(y);
            else return false;
        }
    }

    DVec gather_now(const DVec &src, const DVec &pos) {
        DVec o; o.kind = DVec::DENSE; o.n = pos.n;
        o.data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(o.n, 1));
        o.valid = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(o.n), 1));
        HIP_CHECK(launch_gather(src_of(src), vp(src), src.n, src_of(pos), vp(pos), pos.n, (int64_t *)o.data->p, (uint64_t *)o.valid->p, s));
        note_subset(o.valid, pos.valid);
        return o;
    }
    // FoldSelect statements the planner could turn into a filter over raw table columns (FusedPlan::filters): they
    // read the columns themselves, so the comparison / connective statements feeding them need not run for their sake
    bool filter_on = !getenv("VDL_NO_FILTER_FUSION");
    bool column_filter(const Node &n) const { return filter_on && n.op == Op::FoldSelect && p->fused.filters.count(n.id) != 0; }
    // Gather statements that need not run: read only by `Gather(this, filter)`, or only by the filter idiom
    // FoldSelect(RangeV 0 1 this, this) (the RangeV then only lends its length and an upper bound of the validity)
    std::vector<char> lazy_gather_ok;

    DVec densify(const DVec &v) {
        if (v.kind == DVec::LAZYG) return gather_now(v.lg->src, v.lg->pos);
        if (v.kind == DVec::EXPR && v.sel) return sparse_to_dense(sx_force(v));       // (never reached: only binary() reads such vectors)
        if (v.kind == DVec::EXPR) return expr_force(v);
        if (v.kind == DVec::SPARSE) return sparse_to_dense(v);
        if (v.kind != DVec::ONEHOT && v.kind != DVec::OHCONST) return v;
        DVec src = v;
        if (v.kind == DVec::OHCONST) {   // materialise the constant into a one-hot record first
            BufP oh = dev_alloc(c, 3 * sizeof(int64_t));
            HIP_CHECK(launch_onehot_const(B_MUL, (const int64_t *)v.data->p, 0, 0, (int64_t *)oh->p, s));
            HIP_CHECK(launch_onehot_const(B_ADD, (const int64_t *)oh->p, v.from, 0, (int64_t *)oh->p, s));
            src.data = oh;
        }
        DVec o;
        o.kind = DVec::DENSE; o.n = v.n;
        o.data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(v.n, 1));
        o.valid = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(v.n), 1));
        HIP_CHECK(launch_onehot_dense((const int64_t *)src.data->p, (int64_t *)o.data->p, (uint64_t *)o.valid->p, v.n, s));
        return o;
    }

    bool descends(const SelP &x, const SelP &from) const {
        for (SelP k = x; k; k = k->parent) if (k == from) return true;
        return false;
    }
    // SPARSE op SPARSE (on one selection, or one selection filtered out of the other), or SPARSE op constant
    // whose validity covers the selection
    bool sparse_binary(const Node &n, const DVec &a0, const DVec &b0, DVec &o) {
        DVec a = a0, b = b0;
        if (a.kind == DVec::SPARSE && b.kind == DVec::SPARSE && a.sel != b.sel) {
            DVec t;
            if (descends(b.sel, a.sel) && sparse_narrow(a, b.sel, t)) a = t;
            else if (descends(a.sel, b.sel) && sparse_narrow(b, a.sel, t)) b = t;
            else return false;
        }
        const SelP sel = a.kind == DVec::SPARSE ? a.sel : b.sel;
        auto as_const = [&](const DVec &k, Src &out) {
            if (!(k.kind == DVec::RANGE && k.step == 0)) return false;
            if (k.valid && !subset(bitmap_of(sel), k.valid)) return false;
            out.kind = SRC_RANGE; out.from = k.from; out.step = 0;
            return true;
        };
        Src sa, sb;
        if (a.kind == DVec::SPARSE) sa = i64_src(a.data); else if (!as_const(a, sa)) return false;
        if (b.kind == DVec::SPARSE) sb = i64_src(b.data); else if (!as_const(b, sb)) return false;
        BufP data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(sel->m, 1));
        HIP_CHECK(launch_binary(n.bin, sa, sb, (int64_t *)data->p, sel->m, s));
        o = make_sparse(sel, data);
        return true;
    }

    // ---- element-wise chains over the entries of one selection ---------------------------------------------------
    // The same laziness as expr_binary, for SPARSE operands: a Binary whose only reader is another Binary is kept as a
    // tree over the entry buffers (DVec kind EXPR with `sel` set) and runs as one k_expr over the m entries when the chain
    // ends (Q3's composite key is ten such operators).  Only binary() ever sees such a vector (that is what "only reader is
    // a Binary" guarantees); anything it cannot extend forces it into a plain SPARSE vector first.
    static bool is_sx(const DVec &v) { return v.kind == DVec::EXPR && v.sel; }
    DVec sx_force(const DVec &v) {
        const int64_t m = v.sel->m;
        BufP data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(m, 1));
        if (m > 0) {
            ExprProg prog;
            expr_emit(*v.ex, prog);
            HIP_CHECK(launch_expr(prog, (int64_t *)data->p, m, s));
        }
        return make_sparse(v.sel, data);
    }
    bool sparse_expr_binary(const Node &n, const DVec &a, const DVec &b, DVec &o) {
        if (!fuse_on || n.bin == B_DIV || n.bin == B_MOD || getenv("VDL_NO_SPARSE_EXPR")) return false;
        SelP sel;
        for (const DVec *v : {&a, &b})
            if (v->kind == DVec::SPARSE || is_sx(*v)) { if (sel && v->sel != sel) return false; sel = v->sel; }
        if (!sel || sel->m <= 0) return false;
        auto node_of = [&](const DVec &v, std::shared_ptr<ExprNode> &out) {
            if (is_sx(v)) { out = v.ex; return true; }
            auto e = std::make_shared<ExprNode>();
            if (v.kind == DVec::SPARSE) { e->leaf = entries(v); out = e; return true; }
            if (v.kind == DVec::RANGE && v.step == 0 && (!v.valid || subset(bitmap_of(sel), v.valid))) {
                e->leaf = DVec{}; e->leaf.kind = DVec::RANGE; e->leaf.n = sel->m; e->leaf.from = v.from; e->leaf.step = 0;
                out = e; return true;
            }
            return false;
        };
        std::shared_ptr<ExprNode> el, er;
        if (!node_of(a, el) || !node_of(b, er)) return false;
        // (RangeV readers want the shape of a vector, not its values -- exec(RangeV) serves them from the selection -- so they do not
        // end a chain: Q3's composite key, whose every step also feeds a constant-making RangeV, is ONE kernel over the survivors)
        const bool lazy = value_uses[(size_t)n.id] == 1 && value_read_by_binary_only[(size_t)n.id];
        if (!lazy && el->bin < 0 && er->bin < 0) return false;                // a lone operator: the plain kernel
        auto t = std::make_shared<ExprNode>();
        t->bin = n.bin; t->l = el; t->r = er;
        t->leaves = el->leaves + er->leaves;
        t->instrs = el->instrs + er->instrs + 1;
        t->depth = std::max(el->depth, er->depth + 1);
        if (!expr_fits(*t)) return false;                                      // the caller forces the pending sides and goes operator by operator
        o = DVec{};
        o.kind = DVec::EXPR; o.n = sel->n; o.sel = sel; o.ex = t;
        if (!lazy) o = sx_force(o);
        return true;
    }

    DVec binary(const Node &n, const DVec &a0, const DVec &b0) {
        DVec a = a0, b = b0;
        if (a.kind == DVec::SPARSE || b.kind == DVec::SPARSE || is_sx(a) || is_sx(b)) {
            if (a.n == b.n) {
                DVec o;
                if (sparse_expr_binary(n, a, b, o)) return o;
            }
            if (is_sx(a)) a = sx_force(a);
            if (is_sx(b)) b = sx_force(b);
        }
        if (a.kind == DVec::SPARSE || b.kind == DVec::SPARSE) {
            if (a.n != b.n)
                throw Error(VDL_ERR_SHAPE, std::string(kBinNames[n.bin]) + " (Id " + std::to_string(n.id) + "): operand lengths differ (" +
                                               std::to_string(a.n) + " vs " + std::to_string(b.n) + ")");
            DVec o;
            if (sparse_binary(n, a, b, o)) return o;
            a = densify(a); b = densify(b);
        }
        {
            DVec o;
            if (expr_binary(n, a, b, o)) return o;
            if (a.kind == DVec::EXPR) a = expr_force(a);
            if (b.kind == DVec::EXPR) b = expr_force(b);
        }
        const bool a_oh = a.kind == DVec::ONEHOT || a.kind == DVec::OHCONST;
        const bool b_oh = b.kind == DVec::ONEHOT || b.kind == DVec::OHCONST;
        if (a.n != b.n)
            throw Error(VDL_ERR_SHAPE, std::string(kBinNames[n.bin]) + " (Id " + std::to_string(n.id) + "): operand lengths differ (" +
                                           std::to_string(a.n) + " vs " + std::to_string(b.n) + ")");
        if (a_oh && b_oh) {
            DVec o; o.n = a.n;
            if (a.kind == DVec::OHCONST && b.kind == DVec::OHCONST) {
                if (a.data != b.data) { a = densify(a); b = densify(b); }
                else { o = a; o.from = apply_bin(n.bin, a.from, b.from); return o; }
            } else {
                o.kind = DVec::ONEHOT;
                o.data = dev_alloc(c, 3 * sizeof(int64_t));
                if (a.kind == DVec::ONEHOT && b.kind == DVec::ONEHOT)
                    HIP_CHECK(launch_onehot_binary(n.bin, (const int64_t *)a.data->p, (const int64_t *)b.data->p, (int64_t *)o.data->p, s));
                else if (a.kind == DVec::ONEHOT && a.data == b.data)
                    HIP_CHECK(launch_onehot_const(n.bin, (const int64_t *)a.data->p, b.from, 0, (int64_t *)o.data->p, s));
                else if (b.kind == DVec::ONEHOT && a.data == b.data)
                    HIP_CHECK(launch_onehot_const(n.bin, (const int64_t *)b.data->p, a.from, 1, (int64_t *)o.data->p, s));
                else { a = densify(a); b = densify(b); o.kind = DVec::NONE; }
                if (o.kind == DVec::ONEHOT) return o;
            }
        } else if (a_oh || b_oh) {
            a = densify(a); b = densify(b);
        }
        DVec o;
        o.n = a.n;
        if (a.kind == DVec::RANGE && b.kind == DVec::RANGE && a.step == 0 && b.step == 0) {   // constant folding
            o.kind = DVec::RANGE; o.from = apply_bin(n.bin, a.from, b.from); o.step = 0;
            o.valid = and_valid(a, b, o.n);
            return o;
        }
        o.kind = DVec::DENSE;
        o.data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(o.n, 1));
        HIP_CHECK(launch_binary(n.bin, src_of(a), src_of(b), (int64_t *)o.data->p, o.n, s));
        o.valid = and_valid(a, b, o.n);
        return o;
    }

    // Result transfers of pinned outputs are queued on the context's copy stream behind the kernels that produced them
    // and overlap the statements that follow; the run waits for them at its end.
    std::vector<BufP> copies_in_flight;
    void copy_out(Output &o, const BufP &dev, size_t count) {
        if (p->device_outputs && count >= kBigOutput) {  // the caller reads it where it is
            o.dev_keep = dev; o.dev = (const int64_t *)dev->p; o.big_n = count;
            return;
        }
        int64_t *dst = host_out(o, count);
        if (!o.big) {
            // small: through the context's pinned staging area, in stream order, ONE synchronise for all of them when the run ends
            // (a copy into pageable memory blocks the host until the GPU has caught up: six result columns of a few rows each were
            // six idle gaps of 30-40 us)
            int64_t *stage = c->small_stage();
            if (stage && count <= vdl_ctx::kSmallStageWords) {
                if (stage_used + count > vdl_ctx::kSmallStageWords) flush_small();
                HIP_CHECK(hipMemcpyAsync(stage + stage_used, dev->p, sizeof(int64_t) * count, hipMemcpyDeviceToHost, s));
                small_copies.push_back({p->outs.size(), stage_used, count});
                stage_used += count;
                return;
            }
            HIP_CHECK(hipMemcpyAsync(dst, dev->p, sizeof(int64_t) * count, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            return;
        }
        if (!c->copy_stream) {
            HIP_CHECK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&c->copy_ev, hipEventDisableTiming));
        }
        HIP_CHECK(hipEventRecord(c->copy_ev, s));
        HIP_CHECK(hipStreamWaitEvent(c->copy_stream, c->copy_ev, 0));
        HIP_CHECK(hipMemcpyAsync(dst, dev->p, sizeof(int64_t) * count, hipMemcpyDeviceToHost, c->copy_stream));
        copies_in_flight.push_back(dev);                // the pool must not hand the buffer out again before the copy ran
    }
    struct SmallCopy { size_t out, at, count; };
    std::vector<SmallCopy> small_copies;
    size_t stage_used = 0;
    void flush_small() {
        if (small_copies.empty()) return;
        HIP_CHECK(hipStreamSynchronize(s));
        const int64_t *stage = c->small_stage();
        for (const SmallCopy &k : small_copies) std::memcpy(p->outs[k.out].vals.data(), stage + k.at, sizeof(int64_t) * k.count);
        small_copies.clear();
        stage_used = 0;
    }
    void finish_copies() {
        flush_small();
        if (copies_in_flight.empty()) return;
        HIP_CHECK(hipStreamSynchronize(c->copy_stream));
        copies_in_flight.clear();
    }

    // where the values of an output go on the host: a pinned buffer of the plan when large
    int64_t *host_out(Output &o, size_t count) {
        if (count >= kBigOutput) {
            int64_t *pin = p->pinned_out(p->outs.size(), count);
            if (pin) { o.big = pin; o.big_n = count; return pin; }
        }
        o.vals.resize(count);
        return o.vals.data();
    }

    void materialize(const Node &n, const DVec &v0) {
        Output o;
        o.node = n.id;
        o.name = n.field;
        o.tmp = "tmp" + std::to_string(n.id);
        DVec v = v0;
        if (v.kind == DVec::SPARSE) {                       // every entry holds a value: the output is the entries
            if (v.sel->m > 0) copy_out(o, v.data, (size_t)v.sel->m);
            p->outs.push_back(std::move(o));
            return;
        }
        if (v.kind == DVec::OHCONST) v = densify(v);
        if (v.kind == DVec::ONEHOT) {
            int64_t h[3];
            fetch_words(v.data->p, 3, h);
            if (h[2] > 0) o.vals.push_back(h[0]);
        } else {
            const int64_t nb = (v.n + compact_tile() - 1) / compact_tile();
            if (nb > 0) {
                BufP counts = dev_alloc(c, sizeof(int64_t) * (size_t)(nb + 1));
                HIP_CHECK(launch_compact_offsets(vp(v), v.n, (int64_t *)counts->p, s));
                int64_t total = 0;
                fetch_words((int64_t *)counts->p + nb, 1, &total);
                if (total > 0) {
                    BufP outb = dev_alloc(c, sizeof(int64_t) * (size_t)total);
                    HIP_CHECK(launch_compact_write(src_of(v), vp(v), v.n, (const int64_t *)counts->p, (int64_t *)outb->p, s));
                    copy_out(o, outb, (size_t)total);
                }
            }
        }
        p->outs.push_back(std::move(o));
    }

    // entries of a SPARSE vector on a selection filtered (possibly in several steps) out of its own
    bool sparse_narrow(const DVec &src, const SelP &to, DVec &o) {
        std::vector<SelP> chain;
        for (SelP x = to; x && x != src.sel; x = x->parent) chain.push_back(x);
        if (chain.empty() || chain.back()->parent != src.sel) return false;
        BufP cur = src.data;
        int64_t cur_m = src.sel->m;
        for (size_t k = chain.size(); k-- > 0;) {
            const SelP &step = chain[k];
            BufP d = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(step->m, 1));
            BufP junk = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(step->m), 1));
            HIP_CHECK(launch_gather(i64_src(cur), nullptr, cur_m, i64_src(step->ppos), nullptr, step->m, (int64_t *)d->p, (uint64_t *)junk->p, s));
            cur = d; cur_m = step->m;
        }
        o = make_sparse(to, cur);
        o.ids = src.ids;
        return true;
    }

    // Gather with a sparse side: the filter idiom Gather(x, FoldSelect(..)) producing or narrowing a SPARSE vector,
    // and gathers through sparse positions (FK joins of filtered fact rows).  false = take the general route.
    bool sparse_gather(const DVec &src, const DVec &pos0, DVec &o) {
        DVec pos = pos0;
        if (pos.kind == DVec::RANGE && pos.from == 0 && pos.step == 1 && pos.n != src.n && pos.valid) {
            // the slots' own ids on a selection, read out of a vector of ANOTHER length (TPC-H Q18: the orders in its semi-join set -- a
            // set as long as lineitem -- looked up in the orders table): the selection's slot list is the position vector
            SelP sel = sel_for(pos.valid, pos.n);
            if (sel->worth && sel->idx) { pos = make_sparse(sel, sel->idx); pos.ids = true; }
        }
        const bool identity = pos.kind == DVec::RANGE && pos.from == 0 && pos.step == 1 && pos.n == src.n;
        if (identity) {
            if (!pos.valid) return false;
            if (src.kind == DVec::LAZYG) {
                // Gather(Gather(x, p), filter): only the filter's rows of the inner gather are ever needed
                const LazyGather &lg = *src.lg;
                if (!subset(pos.valid, lg.pos.valid)) return false;
                SelP sel = sel_for(pos.valid, src.n);
                if (!sel->worth) return false;
                DVec pe = sparse_take(lg.pos, sel);                                            // the inner positions on the selection
                BufP data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(sel->m, 1));
                BufP sub = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(sel->m), 1));
                HIP_CHECK(launch_gather(src_of(lg.src), vp(lg.src), lg.src.n, i64_src(pe.data), nullptr, sel->m, (int64_t *)data->p, (uint64_t *)sub->p, s));
                o = sparse_normalised(sel, data, sub);
                return true;
            }
            if (src.kind == DVec::SPARSE) {
                if (subset(bitmap_of(src.sel), pos.valid)) { o = src; return true; }             // the filter keeps every entry
                auto it = sel_of_bitmap.find(pos.valid->p);
                if (it != sel_of_bitmap.end() && sparse_narrow(src, it->second, o)) return true;
                return false;
            }
            if (!(src.kind == DVec::DENSE || src.kind == DVec::COLUMN || src.kind == DVec::RANGE)) return false;
            BufP both = src.valid ? and_bitmaps(src.valid, pos.valid, src.n) : pos.valid;
            SelP sel = sel_for(both, src.n);
            if (!sel->worth) return false;
            if (src.kind == DVec::RANGE && src.step == 0) return false;                        // constants stay virtual
            if (src.kind == DVec::RANGE && src.from == 0 && src.step == 1 && sel->idx) {        // row ids through a filter = the selection's slot list
                o = make_sparse(sel, sel->idx);
                o.ids = true;
                return true;
            }
            o = sparse_take(src, sel);
            return true;
        }
        if (pos.kind == DVec::SPARSE && src.kind == DVec::SPARSE && src.sel->idx && src.sel->m > 0 && !getenv("VDL_NO_RANKED_GATHER")) {
            // both sides sparse (a join reading a filtered dimension column through the fact side's surviving keys): the
            // entry of a source slot is found by rank in the source selection's bitmap; no dense copy of the source
            const SelP &ss = src.sel;
            const BufP &bits = bitmap_of(ss);
            if (!ss->wrank) {
                const int64_t nw = nwords(ss->n);
                ss->wrank = dev_alloc(c, sizeof(int64_t) * (size_t)(nw + 1));
                BufP sums = dev_alloc(c, sizeof(int64_t) * (size_t)(prefix_sum_blocks(nw) + 2));
                HIP_CHECK(launch_word_counts((const uint64_t *)bits->p, nw, (int64_t *)ss->wrank->p, s));
                HIP_CHECK(launch_prefix_sum((int64_t *)ss->wrank->p, nw, (int64_t *)sums->p, s));
            }
            const SelP &sel = pos.sel;
            BufP data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(sel->m, 1));
            BufP sub = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(sel->m), 1));
            HIP_CHECK(launch_gather_ranked((const int64_t *)src.data->p, (const uint64_t *)bits->p, (const int64_t *)ss->wrank->p, src.n,
                                           i64_src(pos.data), nullptr, sel->m, (int64_t *)data->p, (uint64_t *)sub->p, s));
            o = sparse_normalised(sel, data, sub);
            return true;
        }
        if (pos.kind == DVec::SPARSE) {
            DVec from = densify(src);
            if (!(from.kind == DVec::DENSE || from.kind == DVec::COLUMN || from.kind == DVec::RANGE)) return false;
            const SelP &sel = pos.sel;
            BufP data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(sel->m, 1));
            BufP sub = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(sel->m), 1));
            HIP_CHECK(launch_gather(src_of(from), vp(from), from.n, i64_src(pos.data), nullptr, sel->m, (int64_t *)data->p, (uint64_t *)sub->p, s));
            // which entries come back with a value depends on the positions, the source's length and its validity alone: the columns
            // of one table read through one key (a join printing five columns of the dimension) share the answer -- one count and one
            // round trip to the host instead of one per column
            const auto key = std::make_tuple((const void *)pos.data->p, (const void *)(from.valid ? from.valid->p : nullptr), from.n);
            auto known = gather_live.find(key);
            if (known != gather_live.end() && known->second.sel == sel && !getenv("VDL_NO_GATHER_LIVE_CACHE")) {
                const GatherLive &g = known->second;
                o = g.live == sel->m ? make_sparse(sel, data) : make_sparse(g.child, compact_write(i64_src(data), g.sub, sel->m, g.offsets, g.live));
                return true;
            }
            GatherLive g;
            g.pos = pos.data; g.fvalid = from.valid; g.sel = sel; g.sub = sub;
            g.live = popcount(sub, sel->m, &g.offsets);
            if (g.live == sel->m) o = make_sparse(sel, data);
            else {
                g.child = child_selection(sel, sub, g.live, g.offsets);
                o = make_sparse(g.child, compact_write(i64_src(data), sub, sel->m, g.offsets, g.live));
            }
            gather_live[key] = g;
            return true;
        }
        return false;
    }

    // FoldSelect over general runs (oracle/vdl_oracle.c:op_fold F_SEL): inside every run of the control vector (EPS
    // control slots skipped) the slot ids of the non-zero data are packed into the run's first member slots.
    // On the m non-EPS control slots: run number by a prefix sum over the run heads; a stable Partition by
    // (run, not selected) ranks the selected entries of a run first, in order, so rank = the member slot to write.
    DVec fold_select_runs(const DVec &ctl, const DVec &d) {
        const int64_t n = d.n;
        DVec o; o.kind = DVec::DENSE; o.n = n;
        o.data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(n, 1));
        o.valid = zero_bitmap(n);
        BufP offsets;
        const int64_t m = popcount(ctl.valid, n, &offsets);
        if (m == 0) return o;
        BufP idx = compact_write(iota_src(), ctl.valid, n, offsets, m);             // member slots
        auto at_members = [&](const DVec &v, BufP &vals, BufP &vbits) {
            vals = dev_alloc(c, sizeof(int64_t) * (size_t)m);
            vbits = dev_alloc(c, sizeof(uint64_t) * (size_t)nwords(m));
            HIP_CHECK(launch_gather(src_of(v), vp(v), v.n, i64_src(idx), nullptr, m, (int64_t *)vals->p, (uint64_t *)vbits->p, s));
        };
        BufP ce, cv, de, dv;
        at_members(ctl, ce, cv);
        at_members(d, de, dv);
        BufP flags = dev_alloc(c, sizeof(int64_t) * (size_t)m), excl = dev_alloc(c, sizeof(int64_t) * (size_t)m);
        HIP_CHECK(launch_run_heads((const int64_t *)ce->p, m, (int64_t *)flags->p, (int64_t *)excl->p, s));
        BufP sums = dev_alloc(c, sizeof(int64_t) * (size_t)(prefix_sum_blocks(m) + 2));
        HIP_CHECK(launch_prefix_sum((int64_t *)excl->p, m, (int64_t *)sums->p, s));
        BufP keys = dev_alloc(c, sizeof(int64_t) * (size_t)m);
        BufP selected = dev_alloc(c, sizeof(uint64_t) * (size_t)nwords(m));
        HIP_CHECK(launch_fsel_keys((const int64_t *)excl->p, (const int64_t *)flags->p, (const int64_t *)de->p, (const uint64_t *)dv->p, m,
                                   (int64_t *)keys->p, (uint64_t *)selected->p, s));
        int64_t last[2];
        {                                                   // (two words from two buffers: side by side in the pinned words, one wait)
            int64_t *pin = c->pinned(2);
            int64_t *to = pin ? pin : last;
            HIP_CHECK(hipMemcpyAsync(&to[0], (const int64_t *)excl->p + (m - 1), sizeof(int64_t), hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipMemcpyAsync(&to[1], (const int64_t *)flags->p + (m - 1), sizeof(int64_t), hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            if (pin) { last[0] = pin[0]; last[1] = pin[1]; }
        }
        const int64_t nruns = last[0] + last[1];
        DVec kv; kv.kind = DVec::DENSE; kv.n = m; kv.data = keys;
        DVec rank = partition_positions(kv, 0, 2 * nruns);
        BufP target = dev_alloc(c, sizeof(int64_t) * (size_t)m), junk = dev_alloc(c, sizeof(uint64_t) * (size_t)nwords(m));
        HIP_CHECK(launch_gather(i64_src(idx), nullptr, m, i64_src(rank.data), nullptr, m, (int64_t *)target->p, (uint64_t *)junk->p, s));
        HIP_CHECK(launch_scatter(i64_src(idx), (const uint64_t *)selected->p, i64_src(target), nullptr, m, n, (int64_t *)o.data->p, (uint64_t *)o.valid->p, s));
        return o;
    }

    // positions of a Partition that were left in rank order (DVec::order): written out now, for a reader that is not a Scatter
    void need_positions(DVec &v) {
        if (v.data || !(v.order || (v.iota && v.perm))) return;
        const int64_t m = v.kind == DVec::SPARSE ? v.sel->m : v.n;
        v.data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(m, 1));
        if (m <= 0) return;
        if (v.order) HIP_CHECK(launch_scatter(iota_src(), nullptr, i64_src(v.order), nullptr, m, m, (int64_t *)v.data->p, nullptr, s));
        else {                                              // the identity (the partitioned data was in order)
            Src zero; zero.kind = SRC_RANGE; zero.from = 0; zero.step = 0;
            HIP_CHECK(launch_binary(B_ADD, iota_src(), zero, (int64_t *)v.data->p, m, s));
        }
    }
    std::vector<char> order_only;          // Partition statements whose every reader is a Scatter taking them as positions

    // Partition positions of `data` over the pivots RangeC pmin pcount 1 (EPS in -> EPS out)
    DVec partition_positions(const DVec &data, int64_t pmin, int64_t pcount, bool order_only = false) {
        DVec o;
        o.kind = DVec::DENSE; o.n = data.n; o.valid = data.valid;
        o.perm = !data.valid;                                   // every slot gets a rank: a permutation of 0 .. n-1
        o.ranks = true;                                         // in any case the valid slots get the ranks 0 .. m-1
        o.data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(o.n, 1));
        int64_t max_bucket = -1;
        bool values_inside = false;
        if (o.n > 1 && !data.valid && partition_passes(pcount) > 1 && !getenv("VDL_NO_SORTED_SHORTCUT")) {
            // data already in order (lineitems are clustered by order key: the group keys of Q3 / Q18 arrive sorted)?
            // bucket = clamp(data - min) is monotone, so the stable ranks are then 0, 1, 2, ... without a single radix pass
            // (the same pass leaves the run heads: if the data is in order the folds over it need no head pass and no count of
            // their own -- and their number comes back with the verdict, in this one round trip through pinned memory)
            const int64_t nb = (o.n + compact_tile() - 1) / compact_tile();
            BufP counts = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(nb + 4, sorted_heads_counts_words(o.n)));   // tile counts, their total, then {descends, max, min} (and the fused pass's verdict per block)
            int64_t *flag = (int64_t *)counts->p + nb + 1;
            BufP heads = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(o.n), 1));
            int64_t back[4] = {0, 0, 0, 0};
            int64_t *pin = c->pinned(4), *pflag = c->flag_words();
            const Src ds = src_of(data);
            if (pin && pflag && compact_tile() == 4096 && sorted_heads_counted_serves(ds, o.n)) {
                // one launch: heads, their counts per tile, the scan, the verdict posted into pinned memory (polled: no stream synchronise)
                if (!c->sorted_state) {
                    c->sorted_state = dev_alloc(c, (size_t)sorted_heads_state_words() * sizeof(int64_t));
                    HIP_CHECK(launch_sorted_state_init((int64_t *)c->sorted_state->p, s));
                }
                const int64_t seq = ++c->post_seq;
                HIP_CHECK(launch_sorted_heads_counted((const int64_t *)ds.p, o.n, (uint64_t *)heads->p, (int64_t *)counts->p, (int64_t *)c->sorted_state->p, pin, pflag, seq, s));
                c->wait_seq(pflag, seq, s);
                std::memcpy(back, pin, sizeof back);
            } else {
                HIP_CHECK(launch_sorted_heads(ds, o.n, (uint64_t *)heads->p, flag, s));     // (sets the two flag words itself)
                HIP_CHECK(launch_compact_count((const uint64_t *)heads->p, o.n, (int64_t *)counts->p, s));
                HIP_CHECK(launch_compact_scan((int64_t *)counts->p, nb, s));
                c->fetch_to_host((int64_t *)counts->p + nb, 4, back, s);      // (posted into pinned memory and polled: no stream synchronise)
            }
            const int64_t nheads = back[0], seen[2] = {back[1], back[2]};
            // every value inside the pivots: bucket = value - pmin exactly, so the sorted buckets ARE the sorted values
            values_inside = back[3] >= pmin && (uint64_t)back[2] - (uint64_t)pmin <= (uint64_t)pcount && back[2] >= back[3];
            if (!seen[0] && ((data.kind == DVec::DENSE && data.data) || (data.kind == DVec::COLUMN && data.ptr))) {
                // (COLUMN: the key of a sharded Partition's tail lies in the receive buffer, which the plan keeps for the run)
                SortedHeads &sh = sorted_heads[data.kind == DVec::DENSE ? data.data->p : data.ptr];
                sh.key = data.kind == DVec::DENSE ? data.data : data.keep; sh.heads = heads; sh.offsets = counts; sh.count = nheads;
            }
            const int64_t descends = seen[0];
            // the largest bucket that occurs (bucket = clamp(data - min, 0, cnt) is monotone in the data)
            max_bucket = seen[1] <= pmin ? 0 : ((int64_t)((uint64_t)seen[1] - (uint64_t)pmin) < 0 ? pcount : std::min<int64_t>((int64_t)((uint64_t)seen[1] - (uint64_t)pmin), pcount));
            if (!descends) {
                o.iota = true;
                if (order_only) { o.data = nullptr; return o; }      // only Scatters read them, and a Scatter by the identity moves nothing (need_positions writes them out for anything else)
                Src zero; zero.kind = SRC_RANGE; zero.from = 0; zero.step = 0;
                HIP_CHECK(launch_binary(B_ADD, iota_src(), zero, (int64_t *)o.data->p, o.n, s));
                return o;
            }
        }
        if (o.n > 0) {
            const int passes = partition_passes(pcount);
            BufP scr = dev_alloc(c, partition_scratch_bytes(o.n, pcount));
            BufP nvalid = dev_alloc(c, sizeof(int64_t));
            BufP ka, sa, kb, sb;
            if (passes > 1) {
                ka = dev_alloc(c, sizeof(int64_t) * (size_t)o.n); sa = dev_alloc(c, sizeof(int64_t) * (size_t)o.n);
                kb = dev_alloc(c, sizeof(int64_t) * (size_t)o.n); sb = dev_alloc(c, sizeof(int64_t) * (size_t)o.n);
            }
            // Positions that only Scatters read (the GROUP BY idiom: key and values scattered into key order) are never written: the
            // last radix pass leaves the slots in RANK order, stored like any other pass, and the Scatters become gathers through
            // that list.  `pos[slot] = rank` is 60 M isolated 8-byte stores at 60 M rows -- 1.6 of the Partition's 3.9 ms.
            const bool lazy = order_only && !data.valid && passes > 1;
            if (lazy) o.order = dev_alloc(c, sizeof(int64_t) * (size_t)o.n);
            // ... and the values come out of the sort in rank order for 8 more bytes per row of sequential stores: the Scatter of the key
            // itself by these positions -- every GROUP BY has one -- then costs nothing (it was a 1.2 ms gather at 60 M rows)
            if (lazy && values_inside && data.kind == DVec::DENSE && data.data) {
                o.sorted_keys = dev_alloc(c, sizeof(int64_t) * (size_t)o.n);
                o.sorted_keys_src = data.data;
            }
            HIP_CHECK(launch_partition(src_of(data), vp(data), o.n, pmin, pcount, scr->p,
                                       ka ? (uint64_t *)ka->p : nullptr, sa ? (int64_t *)sa->p : nullptr,
                                       kb ? (uint64_t *)kb->p : nullptr, sb ? (int64_t *)sb->p : nullptr,
                                       (int64_t *)nvalid->p, (int64_t *)o.data->p, s, max_bucket, lazy ? (int64_t *)o.order->p : nullptr,
                                       o.sorted_keys ? (int64_t *)o.sorted_keys->p : nullptr));
            if (lazy) o.data = nullptr;                     // (not written: need_positions() fills it in if anybody asks)
        }
        return o;
    }

    // Fold `n` over m ENTRIES (every one holds a value, entry 0 starts the first run): control values at ctl, this fold's data at
    // dsrc.  sel = the selection the entries sit on (a prefix selection for vectors scattered into key order, or all n slots).
    // Results: SPARSE on the child selection of the run heads, packed.  Every fold of one GROUP BY shares the heads; with the
    // batch on they also share ONE launch (launch_group_fold) -- the sibling folds' operands are run ahead of their turn.
    DVec fold_entries(const Node &n, const SelP &sel, Src ctl_src, const void *ctl_key, const BufP &ctl_keep, Src dsrc, int64_t m, int64_t nslots) {
        auto V = [&](int id) -> const DVec & { return vec[(size_t)id]; };
        const int kind = n.op == Op::FoldSum ? 0 : n.op == Op::FoldMin ? 1 : n.op == Op::FoldMax ? 2 : n.op == Op::FoldCount ? 3 : 4;
        // every entry holds a datum, so each run yields a result at its head: the run heads of this control
        // vector -- and the selection they form -- are computed once and shared by all folds over it
        // (a GROUP BY folds every aggregate over the same sorted key, Vlite.hs:1056-1060)
        RunHeads &rh = heads_of[std::make_pair(ctl_key, (const void *)sel.get())];
        const size_t nw = (size_t)std::max<int64_t>(nwords(m), 1);
        if (!rh.heads) {
            rh.ctl = ctl_keep; rh.sel = sel;
            auto known = sorted_heads.find(ctl_key);
            if (group_batch_on && known != sorted_heads.end()) {
                // the Partition found this key in order and left its run heads and their count (partition_positions)
                rh.heads = known->second.heads; rh.offsets = known->second.offsets; rh.count = known->second.count;
            } else {
                rh.heads = dev_alloc(c, sizeof(uint64_t) * nw);
                rh.wordhd = dev_alloc(c, sizeof(int64_t) * (nw + (size_t)maxscan_blocks((int64_t)nw) + 1));
                HIP_CHECK(launch_fold_heads(ctl_src, nullptr, m, (uint64_t *)rh.heads->p, (int64_t *)rh.wordhd->p, s));
                rh.count = popcount(rh.heads, m, &rh.offsets);
            }
            rh.child = child_selection(sel, rh.heads, rh.count, rh.offsets);
        }
        // one launch for all folds pays when runs are short (a partial per run and word, atomics where a run crosses words): with a
        // handful of long runs every word would add to the same few addresses -- k_seg_fold, which carries a run along a wave's
        // words, is the kernel for those
        if (group_batch_on && (rh.count * 64 >= m || m <= (1 << 16))) {
            std::vector<int> members{n.id};
            std::vector<Src> srcs{dsrc};
            for (int id : p->prog.order) {
                const Node &f = p->prog.at(id);
                if (id == n.id || f.a != n.a || (size_t)id >= needed_now.size() || !needed_now[(size_t)id] || done[(size_t)id]) continue;
                if (f.op != Op::FoldSum && f.op != Op::FoldMin && f.op != Op::FoldMax && f.op != Op::FoldCount && f.op != Op::FoldChoose) continue;
                if (cur_over && cur_over->count(id)) continue;
                ensure(f.b);
                DVec fd = V(f.b);
                if (!sel->idx && sel->m == sel->n && fd.kind == DVec::EXPR && !fd.sel) { fd = expr_force(fd); vec[(size_t)f.b] = fd; }
                if (fd.kind == DVec::SPARSE && fd.sel == sel) srcs.push_back(i64_src(fd.data));
                else if (fd.kind == DVec::RANGE && fd.step == 0 && fd.n == nslots && (sel->m == sel->n ? !fd.valid : subset(bitmap_of(sel), fd.valid))) srcs.push_back(src_of(fd));
                else if (!sel->idx && sel->m == sel->n && (fd.kind == DVec::DENSE || fd.kind == DVec::COLUMN || fd.kind == DVec::RANGE) && !fd.valid && fd.n == nslots) srcs.push_back(src_of(fd));
                else continue;
                members.push_back(id);
            }
            std::vector<BufP> outs(members.size());
            const int64_t G = rh.count;
            for (size_t at = 0; at < members.size(); at += kMaxGroupFolds) {
                GroupFoldArgs ga;
                for (size_t k = at; k < members.size() && k < at + kMaxGroupFolds; k++) {
                    const Op fop = p->prog.at(members[k]).op;
                    const int fk = fop == Op::FoldSum ? 0 : fop == Op::FoldMin ? 1 : fop == Op::FoldMax ? 2 : fop == Op::FoldCount ? 3 : 4;
                    outs[k] = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(G, 1));
                    if (fk != 4 && G > 0)
                        HIP_CHECK(launch_fill_words((uint64_t *)outs[k]->p, fk == 1 ? (uint64_t)INT64_MAX : fk == 2 ? (uint64_t)INT64_MIN : 0ull, G, s));
                    ga.kind[ga.nfold] = fk; ga.data[ga.nfold] = srcs[k]; ga.out[ga.nfold] = (int64_t *)outs[k]->p;
                    ga.nfold++;
                }
                if (G > 0) HIP_CHECK(launch_group_fold(ga, (const uint64_t *)rh.heads->p, m, (const int64_t *)rh.offsets->p, s));
            }
            for (size_t k = 1; k < members.size(); k++) {
                vec[(size_t)members[k]] = make_sparse(rh.child, outs[k]);
                done[(size_t)members[k]] = 1;
                if (p->tracing) snapshot(p->prog.at(members[k]), vec[(size_t)members[k]]);
            }
            return make_sparse(rh.child, outs[0]);
        }
        if (!rh.wordhd) {                                   // (heads adopted from the Partition: the per-word lookup k_seg_fold wants)
            rh.wordhd = dev_alloc(c, sizeof(int64_t) * (nw + (size_t)maxscan_blocks((int64_t)nw) + 1));
            HIP_CHECK(launch_fold_heads(ctl_src, nullptr, m, (uint64_t *)rh.heads->p, (int64_t *)rh.wordhd->p, s));
        }
        // FoldChoose takes a run's first element, and here every entry holds one: the values at the run heads, i.e. one
        // compaction by the head bitmap (every output column of a GROUP BY is such a fold, Vlite.hs:1056-1060)
        if (kind == 4) return make_sparse(rh.child, compact_write(dsrc, rh.heads, m, rh.offsets, rh.count));
        BufP data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(m, 1));
        BufP vout = zero_bitmap(m);
        HIP_CHECK(launch_fold_runs(kind, dsrc, nullptr, nullptr, (const uint64_t *)rh.heads->p, (const int64_t *)rh.wordhd->p, m,
                                   (int64_t *)data->p, (uint64_t *)vout->p, s));
        return make_sparse(rh.child, compact_write(i64_src(data), rh.heads, m, rh.offsets, rh.count));
    }

    // a statement (and what it depends on) ahead of its turn: statements are pure, so order does not matter; the main loop skips it
    void ensure(int id) {
        if (id <= 0 || done[(size_t)id] || vec[(size_t)id].kind != DVec::NONE) return;
        if (cur_over && cur_over->count(id)) { vec[(size_t)id] = cur_over->at(id); done[(size_t)id] = 1; return; }
        const Node &n = p->prog.at(id);
        if (!column_filter(n))
            for (int opnd : {n.a, n.b, n.c}) ensure(opnd);
        vec[(size_t)id] = exec(n);
        done[(size_t)id] = 1;
        if (p->tracing) snapshot(n, vec[(size_t)id]);
    }

    DVec exec(const Node &n) {
        auto V = [&](int id) -> const DVec & { return vec[(size_t)id]; };
        DVec o;
        switch (n.op) {
        case Op::Load: {
            const Column &col = find_col(c, n.column);
            o.kind = DVec::COLUMN; o.n = col.n; o.ptr = col.dev; o.width = col.width; o.keep = col.owned;
            return o;
        }
        case Op::Project: case Op::Shuffle:
            return V(n.a);
        case Op::RangeV: {
            const DVec &r = V(n.a);
            if (r.kind == DVec::ONEHOT || r.kind == DVec::OHCONST) {
                if (n.imm1 == 0) { o.kind = DVec::OHCONST; o.n = r.n; o.data = r.data; o.from = n.imm0; return o; }
                // only WHERE the record's one value sits matters to a range over it: its validity, not its n-slot dense form (TPC-H Q11:
                // 110 us of 840 went into writing 8 M zeros for the sake of one bit)
                o.kind = DVec::RANGE; o.n = r.n; o.from = n.imm0; o.step = n.imm1;
                o.valid = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(r.n), 1));
                HIP_CHECK(launch_onehot_dense((const int64_t *)r.data->p, nullptr, (uint64_t *)o.valid->p, r.n, s));
                return o;
            }
            if (r.kind == DVec::EXPR && r.sel) {               // a pending chain over the entries of a selection: values exactly there
                o.kind = DVec::RANGE; o.n = r.n; o.from = n.imm0; o.step = n.imm1; o.valid = bitmap_of(r.sel);
                return o;
            }
            if (r.kind == DVec::EXPR) {
                o.kind = DVec::RANGE; o.n = r.n; o.from = n.imm0; o.step = n.imm1; o.valid = expr_valid(*r.ex, r.n);
                return o;
            }
            if (r.kind == DVec::LAZYG) {          // only as the control of FoldSelect over the same gather (lazy_gather_ok): an upper bound will do
                o.kind = DVec::RANGE; o.n = r.n; o.from = n.imm0; o.step = n.imm1; o.valid = r.lg->pos.valid;
                return o;
            }
            if (r.kind == DVec::SPARSE) {
                o.kind = DVec::RANGE; o.n = r.n; o.from = n.imm0; o.step = n.imm1; o.valid = bitmap_of(r.sel);
                return o;
            }
            o.kind = DVec::RANGE; o.n = r.n; o.from = n.imm0; o.step = n.imm1; o.valid = r.valid;
            return o;
        }
        case Op::RangeC:
            o.kind = DVec::RANGE; o.n = n.imm1; o.from = n.imm0; o.step = n.imm2;
            return o;
        case Op::Binary:
            return binary(n, V(n.a), V(n.b));
        case Op::FoldSelect: {
            if (column_filter(n)) {
                const FilterSpec &fs = p->fused.filters.at(n.id);
                FilterArgs fa;
                fa.ncol = (int)fs.cols.size(); fa.never = fs.never ? 1 : 0;
                int64_t rows = -1;
                for (int k = 0; k < fa.ncol; k++) {
                    const Column &col = find_col(c, fs.cols[(size_t)k].name);
                    if (rows >= 0 && col.n != rows)
                        throw Error(VDL_ERR_SHAPE, "columns of table '" + fs.table + "' have different lengths in the catalog");
                    rows = col.n;
                    fa.col[k].p = col.dev;
                    fa.col[k].kind = col.width == 8 ? SRC_I64 : col.width == 4 ? SRC_I32 : col.width == 2 ? SRC_I16 : SRC_I8;
                    fa.nint[k] = fs.cols[(size_t)k].n;
                    for (int j = 0; j < fa.nint[k]; j++) { fa.lo[k][j] = fs.cols[(size_t)k].lo[j]; fa.hi[k][j] = fs.cols[(size_t)k].hi[j]; }
                }
                o.kind = DVec::RANGE; o.n = rows; o.from = 0; o.step = 1;
                o.valid = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(rows), 1));
                HIP_CHECK(launch_filter_columns(fa, (uint64_t *)o.valid->p, rows, s));
                return o;
            }
            if (V(n.b).kind == DVec::SPARSE) {
                // filter of a filtered vector: the new selection is carved out of the entries, not out of the n slots
                const DVec &sd = V(n.b);
                DVec ctl0 = densify(V(n.a));
                if (ctl0.n == sd.n && ctl0.kind == DVec::RANGE && ctl0.step != 0 && subset(bitmap_of(sd.sel), ctl0.valid)) {
                    const SelP &ps = sd.sel;
                    BufP flags = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(ps->m), 1));
                    HIP_CHECK(launch_select_bitmap(i64_src(sd.data), nullptr, nullptr, (uint64_t *)flags->p, ps->m, s));
                    BufP offsets;
                    const int64_t count = popcount(flags, ps->m, &offsets);
                    SelP ch = child_selection(ps, flags, count, offsets);
                    o.kind = DVec::RANGE; o.n = sd.n; o.from = 0; o.step = 1; o.valid = bitmap_of(ch);
                    return o;
                }
            }
            if (V(n.b).kind == DVec::EXPR && pred_on) {
                DVec ctl0 = densify(V(n.a));
                PredProg prog;
                if (ctl0.n == V(n.b).n && ctl0.kind == DVec::RANGE && ctl0.step != 0 && pred_emit(*V(n.b).ex, prog, 0)) {
                    BufP ok = expr_valid(*V(n.b).ex, ctl0.n);
                    if (ctl0.valid && ctl0.valid != ok) ok = ok ? and_bitmaps(ok, ctl0.valid, ctl0.n) : ctl0.valid;
                    o.kind = DVec::RANGE; o.n = ctl0.n; o.from = 0; o.step = 1;
                    o.valid = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(o.n), 1));
                    HIP_CHECK(launch_pred(prog, ok ? (const uint64_t *)ok->p : nullptr, (uint64_t *)o.valid->p, o.n, s));
                    note_subset(o.valid, ok);
                    return o;
                }
            }
            if (V(n.b).kind == DVec::LAZYG) {
                DVec ctl0 = densify(V(n.a));
                const LazyGather &lg = *V(n.b).lg;
                if (ctl0.n == V(n.b).n && ctl0.kind == DVec::RANGE && ctl0.step != 0) {
                    o.kind = DVec::RANGE; o.n = ctl0.n; o.from = 0; o.step = 1;
                    o.valid = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(o.n), 1));
                    HIP_CHECK(launch_select_gather(src_of(lg.src), vp(lg.src), lg.src.n, src_of(lg.pos), vp(lg.pos), vp(ctl0), (uint64_t *)o.valid->p, o.n, s));
                    note_subset(o.valid, lg.pos.valid); note_subset(o.valid, ctl0.valid);
                    return o;
                }
            }
            DVec ctl = densify(V(n.a)), d = densify(V(n.b));
            if (ctl.n != d.n) throw Error(VDL_ERR_SHAPE, "FoldSelect (Id " + std::to_string(n.id) + "): operand lengths differ");
            if (!(ctl.kind == DVec::RANGE && ctl.step != 0)) return fold_select_runs(ctl, d);
            o.kind = DVec::RANGE; o.n = d.n; o.from = 0; o.step = 1;
            if (d.kind == DVec::RANGE && d.step == 0 && d.from != 0 && (d.valid || ctl.valid)) {
                // a non-zero constant: selected wherever it (and the control) holds a value -- the members of a set kept as its bitmap
                o.valid = d.valid && ctl.valid ? and_bitmaps(d.valid, ctl.valid, d.n) : (d.valid ? d.valid : ctl.valid);
                return o;
            }
            o.valid = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(d.n), 1));
            HIP_CHECK(launch_select_bitmap(src_of(d), vp(d), vp(ctl), (uint64_t *)o.valid->p, d.n, s));
            note_subset(o.valid, d.valid); note_subset(o.valid, ctl.valid);
            return o;
        }
        case Op::Gather: {
            {
                // a scalar read back at position 0 for every slot of another vector -- how the compiler compares a column with the result of
                // an ungrouped aggregate (Gather(result, zeros_ other), Vlite.hs:693-712; TPC-H Q15's `total = (select max(total) ..)`, Q11's
                // HAVING threshold): the scalar as a constant on the other vector's validity.  One round trip for the record instead of a
                // gather over every slot and everything behind it going dense (Q15 at SF10: 62_Gather 248 + 65_FoldSelect 239 us).
                const DVec &sv = V(n.a), &pv = V(n.b);
                if ((sv.kind == DVec::ONEHOT || sv.kind == DVec::OHCONST) && pv.kind == DVec::RANGE && pv.step == 0 && pv.from == 0 && sv.data && !getenv("VDL_NO_SCALAR_BROADCAST")) {
                    int64_t rec[3] = {0, 0, 0};
                    fetch_words(sv.data->p, 3, rec);
                    o.kind = DVec::RANGE; o.n = pv.n; o.step = 0;
                    if (rec[2] > 0 && rec[1] == 0 && sv.n > 0) { o.from = sv.kind == DVec::OHCONST ? sv.from : rec[0]; o.valid = pv.valid; }
                    else { o.from = 0; o.valid = zero_bitmap(pv.n); }          // nothing at slot 0: every slot is empty
                    return o;
                }
            }
            if (sparse_on) {
                DVec fast;
                if (sparse_gather(V(n.a), V(n.b), fast)) return fast;
            }
            DVec src = densify(V(n.a)), pos = densify(V(n.b));
            if (pos.kind == DVec::RANGE && pos.from == 0 && pos.step == 1 && pos.n == src.n) {
                // positions are the slot ids themselves (a filter): a view, no data movement
                o = src;
                o.valid = and_valid(src, pos, src.n);
                if (o.valid != src.valid) o.perm = o.ranks = o.iota = false;     // fewer slots hold a value: no longer "the ranks 0 .. m-1"
                return o;
            }
            // a gather whose only reader is a filter (FoldSelect over it, or Gather of it through a filter) is not run:
            // the reader evaluates it where it needs it
            if (fuse_on && lazy_gather_ok[(size_t)n.id]) {
                o.kind = DVec::LAZYG; o.n = pos.n;
                o.lg = std::make_shared<LazyGather>();
                o.lg->src = src; o.lg->pos = pos;
                return o;
            }
            return gather_now(src, pos);
        }
        case Op::Scatter: {
            {
                // positions that are the slots' own ids (the dim side of a join scatters ones / row ids back by
                // Gather(rowids, FoldSelect(..)), Vlite.hs:1268-1275): the result is the source restricted to those slots
                const DVec &ps = V(n.c), &sv = V(n.a);
                const int64_t nout = V(n.b).n;
                BufP where;
                bool identity = false;
                if (ps.kind == DVec::RANGE && ps.from == 0 && ps.step == 1 && ps.n == nout) { identity = true; where = ps.valid; }
                else if (sparse_on && ps.kind == DVec::SPARSE && ps.ids && ps.n == nout) { identity = true; where = bitmap_of(ps.sel); }
                if (identity && sv.n == ps.n) {
                    if (sv.kind == DVec::RANGE || sv.kind == DVec::DENSE || sv.kind == DVec::COLUMN) {
                        o = sv;
                        o.valid = sv.valid && where ? and_bitmaps(sv.valid, where, nout) : (sv.valid ? sv.valid : where);
                        if (o.valid != sv.valid) o.perm = o.ranks = o.iota = false;
                        return o;
                    }
                    if (sv.kind == DVec::SPARSE && ps.kind == DVec::SPARSE && sv.sel == ps.sel) return sv;
                }
            }
            if (sparse_on && V(n.c).kind == DVec::SPARSE) {
                // positions known only on a selection: m writes instead of n
                const DVec &sp = V(n.c);
                DVec sv = V(n.a);
                const int64_t nout = V(n.b).n;
                bool ok = sv.n == sp.n;
                if (ok && sv.kind == DVec::SPARSE && sv.sel != sp.sel) {
                    DVec t;
                    if (descends(sp.sel, sv.sel) && sparse_narrow(sv, sp.sel, t)) sv = t; else ok = false;
                }
                Src ssrc;
                if (ok && sv.kind == DVec::SPARSE) ssrc = i64_src(sv.data);
                else if (ok && sv.kind == DVec::RANGE && sv.step == 0 && subset(bitmap_of(sp.sel), sv.valid)) { ssrc.kind = SRC_RANGE; ssrc.from = sv.from; ssrc.step = 0; }
                else if (ok && sv.kind == DVec::RANGE && sv.from == 0 && sv.step == 1 && subset(bitmap_of(sp.sel), sv.valid)) ssrc = idx_src(*sp.sel);   // row ids
                else ok = false;
                if (ok && ssrc.kind == SRC_RANGE && ssrc.step == 0 && !sp.perm && !sp.valid) {
                    // a constant at a few positions (a semi-join set: ones at the keys of the groups that pass): the set is its bitmap
                    need_positions(vec[(size_t)n.c]);
                    o.kind = DVec::RANGE; o.n = nout; o.from = ssrc.from; o.step = 0;
                    o.valid = zero_bitmap(nout);
                    HIP_CHECK(launch_set_bits((const int64_t *)sp.data->p, sp.sel->m, (uint64_t *)o.valid->p, s, nout));
                    return o;
                }
                if (ok) {
                    const int64_t m = sp.sel->m;
                    if (sp.perm && m <= nout) {
                        // the positions are a permutation of 0 .. m-1 (Partition): the result lives on the prefix selection
                        SelP pre = prefix_selection(nout, m);
                        if (sp.iota && sv.kind == DVec::SPARSE) return make_sparse(pre, sv.data);      // ... and in entry order: nothing moves
                        if (sp.order && sp.sorted_keys && sv.kind == DVec::SPARSE && sv.data == sp.sorted_keys_src) return make_sparse(pre, sp.sorted_keys);   // the key itself: the sort wrote it
                        BufP data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(m, 1));
                        if (sp.order && !sp.data) {                    // positions left in rank order: out[r] = src[order[r]], stored in sequence
                            BufP junk = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(m), 1));
                            HIP_CHECK(launch_gather(ssrc, nullptr, m, i64_src(sp.order), nullptr, m, (int64_t *)data->p, (uint64_t *)junk->p, s));
                        } else {
                            need_positions(vec[(size_t)n.c]);
                            HIP_CHECK(launch_scatter(ssrc, nullptr, i64_src(sp.data), nullptr, m, m, (int64_t *)data->p, nullptr, s));
                        }
                        return make_sparse(pre, data);
                    }
                    need_positions(vec[(size_t)n.c]);
                    o.kind = DVec::DENSE; o.n = nout;
                    o.data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(nout, 1));
                    o.valid = zero_bitmap(nout);
                    HIP_CHECK(launch_scatter(ssrc, nullptr, i64_src(sp.data), nullptr, m, nout, (int64_t *)o.data->p, (uint64_t *)o.valid->p, s));
                    return o;
                }
            }
            if (!V(n.c).data && V(n.c).iota && V(n.c).perm && !V(n.c).order) {
                // identity positions that were never written: a Scatter of a whole vector by them is that vector; anything else asks for them
                const DVec &lp = V(n.c);
                if (lp.kind == DVec::DENSE && !lp.valid && lp.n == V(n.b).n) {
                    DVec sv = densify(V(n.a));
                    if (sv.n == lp.n) return sv;
                }
                need_positions(vec[(size_t)n.c]);
            }
            if (V(n.c).order && !V(n.c).data) {
                const DVec &lp = V(n.c);
                const bool whole = lp.kind == DVec::DENSE && lp.perm && !lp.valid && lp.n == V(n.b).n;
                DVec sv = whole ? densify(V(n.a)) : DVec{};
                if (whole && !sv.valid && sv.n == lp.n && (sv.kind == DVec::DENSE || sv.kind == DVec::COLUMN || sv.kind == DVec::RANGE)) {
                    // a permutation of all slots, left in rank order: out[r] = src[order[r]]
                    o.kind = DVec::DENSE; o.n = lp.n;
                    if (lp.sorted_keys && sv.kind == DVec::DENSE && sv.data == lp.sorted_keys_src) { o.data = lp.sorted_keys; return o; }      // the key itself: the sort wrote it
                    o.data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(o.n, 1));
                    BufP junk = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(o.n), 1));
                    HIP_CHECK(launch_gather(src_of(sv), nullptr, sv.n, i64_src(lp.order), nullptr, o.n, (int64_t *)o.data->p, (uint64_t *)junk->p, s));
                    return o;
                }
                need_positions(vec[(size_t)n.c]);
            }
            DVec src = densify(V(n.a)), pos = densify(V(n.c));
            const DVec &fold = V(n.b);
            if (src.n != pos.n) throw Error(VDL_ERR_SHAPE, "Scatter (Id " + std::to_string(n.id) + "): source and position lengths differ");
            o.kind = DVec::DENSE; o.n = fold.n;
            o.data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(o.n, 1));
            if (pos.perm && pos.iota && !pos.valid && src.n == o.n) return src;   // the identity (the partitioned data was in order): nothing moves
            if (pos.perm && !pos.valid && !src.valid && src.n == o.n) {      // a permutation of all slots: every slot is written
                HIP_CHECK(launch_scatter(src_of(src), nullptr, src_of(pos), nullptr, src.n, o.n, (int64_t *)o.data->p, nullptr, s));
                return o;
            }
            if (pos.ranks && pos.valid && subset(pos.valid, src.valid) && !getenv("VDL_NO_RANK_SCATTER")) {
                // Partition ranks of a filtered vector (a filter too dense to go sparse: Q1 keeps 98 % of lineitem): the slots
                // written are exactly 0 .. m-1, so the validity of the result is a prefix -- no atomic per element (59 M
                // atomics into 0.9 M bitmap words were 85 % of this kernel)
                const int64_t m = sel_for(pos.valid, pos.n)->m;
                if (m <= o.n) {
                    o.valid = bitmap_of(prefix_selection(o.n, m));
                    HIP_CHECK(launch_scatter(src_of(src), nullptr, src_of(pos), vp(pos), src.n, o.n, (int64_t *)o.data->p, nullptr, s));
                    return o;
                }
            }
            o.valid = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(o.n), 1));
            HIP_CHECK(launch_fill_words((uint64_t *)o.valid->p, 0, nwords(o.n), s));
            HIP_CHECK(launch_scatter(src_of(src), vp(src), src_of(pos), vp(pos), src.n, o.n, (int64_t *)o.data->p, (uint64_t *)o.valid->p, s));
            return o;
        }
        case Op::FoldSum: case Op::FoldMin: case Op::FoldMax: case Op::FoldCount: case Op::FoldChoose: {
            if (sparse_on && V(n.b).kind == DVec::SPARSE && V(n.a).kind == DVec::RANGE && V(n.a).step == 0 && V(n.a).n == V(n.b).n &&
                V(n.b).sel->m > 0 && V(n.b).sel->idx && V(n.a).valid && V(n.a).valid == V(n.b).sel->bitmap) {
                // one run over exactly the selected slots (the ungrouped aggregate of a filtered table, Vlite.hs:636-639):
                // fold the entries; the result sits at the run's first slot = the selection's first slot
                const DVec &sd = V(n.b);
                const int kind = n.op == Op::FoldSum ? 0 : n.op == Op::FoldMin ? 1 : n.op == Op::FoldMax ? 2 : n.op == Op::FoldCount ? 3 : 4;
                BufP scratch = dev_alloc(c, sizeof(int64_t) * 3 * (size_t)fold_scratch_blocks());
                o.kind = DVec::ONEHOT; o.n = sd.n;
                o.data = dev_alloc(c, 3 * sizeof(int64_t));
                // (the record's slot is 0: the run starts at slot 0 of the vector, not at the selection's first slot)
                HIP_CHECK(launch_fold_global(kind, i64_src(sd.data), nullptr, nullptr, sd.sel->m, (int64_t *)scratch->p, (int64_t *)o.data->p, s));
                return o;
            }
            const bool data_on_sel = V(n.a).kind == DVec::SPARSE &&
                                     ((V(n.b).kind == DVec::SPARSE && V(n.a).sel == V(n.b).sel) ||
                                      (V(n.b).kind == DVec::RANGE && V(n.b).step == 0 && V(n.b).n == V(n.a).n && subset(bitmap_of(V(n.a).sel), V(n.b).valid)));
            // (the first run's result belongs in slot 0 of the vector: on the entries that is only the case when the selection
            // starts at slot 0 -- always so for the prefix selections GROUP BY folds over; anything else takes the dense route)
            if (sparse_on && data_on_sel && first_slot_of(V(n.a).sel) == 0) {
                // runs skip EPS slots, so folding the m entries gives the same runs; results sit at run-first entries
                const DVec &sc = V(n.a), &sd = V(n.b);
                const SelP sel = sc.sel;
                Src dsrc = sd.kind == DVec::SPARSE ? i64_src(sd.data) : src_of(sd);
                return fold_entries(n, sel, i64_src(sc.data), sc.data->p, sc.data, dsrc, sel->m, sc.n);
            }
            // The same for a control vector that holds a value in EVERY slot (the key of a GROUP BY over an unfiltered table: Q18
            // groups all lineitems by order): its n slots are the entries, the results live on the selection of the run heads.
            // Only with the batch on and many short runs (fold_entries decides): a handful of long runs is what k_seg_fold is for.
            if (sparse_on && group_batch_on && V(n.a).n > 0 && (V(n.a).kind == DVec::DENSE || V(n.a).kind == DVec::COLUMN) && !V(n.a).valid) {
                const DVec &dc = V(n.a);
                DVec dd = V(n.b);
                if (dd.kind == DVec::EXPR && !dd.sel) dd = expr_force(dd);
                const bool plain = (dd.kind == DVec::DENSE || dd.kind == DVec::COLUMN || dd.kind == DVec::RANGE) && !dd.valid && dd.n == dc.n;   // (all n slots are entries: a RANGE counts along them)
                if (plain && sorted_heads.count(dc.kind == DVec::DENSE ? dc.data->p : dc.ptr)) {      // (the Partition saw it in order and counted its runs)
                    const SortedHeads &sh = sorted_heads[dc.kind == DVec::DENSE ? dc.data->p : dc.ptr];
                    if (sh.count * 64 >= dc.n)
                        return fold_entries(n, prefix_selection(dc.n, dc.n), src_of(dc), dc.kind == DVec::DENSE ? dc.data->p : dc.ptr, dc.data, src_of(dd), dc.n, dc.n);
                }
            }
            DVec ctl = densify(V(n.a)), d = densify(V(n.b));
            if (ctl.n != d.n) throw Error(VDL_ERR_SHAPE, std::string(op_name(n.op, -1)) + " (Id " + std::to_string(n.id) + "): operand lengths differ");
            const int kind = n.op == Op::FoldSum ? 0 : n.op == Op::FoldMin ? 1 : n.op == Op::FoldMax ? 2 : n.op == Op::FoldCount ? 3 : 4;
            if (!(ctl.kind == DVec::RANGE && ctl.step == 0)) {
                // general control vector (grouped aggregates fold data scattered into key order)
                const size_t nw = (size_t)std::max<int64_t>(nwords(d.n), 1);
                o.kind = DVec::DENSE; o.n = d.n;
                o.data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(d.n, 1));
                o.valid = dev_alloc(c, sizeof(uint64_t) * nw);
                HIP_CHECK(launch_fill_words((uint64_t *)o.valid->p, 0, nwords(d.n), s));
                // the folds of one GROUP BY all run over the same sorted key: its run heads are computed once
                const void *kd = ctl.kind == DVec::DENSE && ctl.data ? ctl.data->p : nullptr;
                if (kd) {
                    DenseHeads &dh = dense_heads[std::make_pair(kd, (const void *)(ctl.valid ? ctl.valid->p : nullptr))];
                    if (!dh.heads) {
                        dh.ctl = ctl.data; dh.ctlv = ctl.valid;                 // keep the key's addresses from being reused
                        dh.heads = dev_alloc(c, sizeof(uint64_t) * nw);
                        dh.wordhd = dev_alloc(c, sizeof(int64_t) * (nw + (size_t)maxscan_blocks((int64_t)nw) + 1));
                        HIP_CHECK(launch_fold_heads(src_of(ctl), vp(ctl), d.n, (uint64_t *)dh.heads->p, (int64_t *)dh.wordhd->p, s));
                    }
                    HIP_CHECK(launch_fold_runs(kind, src_of(d), vp(d), vp(ctl), (const uint64_t *)dh.heads->p, (const int64_t *)dh.wordhd->p, d.n,
                                               (int64_t *)o.data->p, (uint64_t *)o.valid->p, s));
                    return o;
                }
                BufP heads = dev_alloc(c, sizeof(uint64_t) * nw);
                BufP wordhd = dev_alloc(c, sizeof(int64_t) * (nw + (size_t)maxscan_blocks((int64_t)nw) + 1));
                HIP_CHECK(launch_fold_segmented(kind, src_of(ctl), vp(ctl), src_of(d), vp(d), d.n, (uint64_t *)heads->p,
                                                (int64_t *)wordhd->p, (int64_t *)o.data->p, (uint64_t *)o.valid->p, s));
                return o;
            }
            BufP scratch = dev_alloc(c, sizeof(int64_t) * 3 * (size_t)fold_scratch_blocks());
            o.kind = DVec::ONEHOT; o.n = d.n;
            o.data = dev_alloc(c, 3 * sizeof(int64_t));
            HIP_CHECK(launch_fold_global(kind, src_of(d), vp(d), vp(ctl), d.n, (int64_t *)scratch->p, (int64_t *)o.data->p, s));
            return o;
        }
        case Op::Partition: {
            if (sparse_on && V(n.a).kind == DVec::SPARSE) {
                const DVec &sd = V(n.a);
                const DVec &pv = V(n.b);
                if (pv.kind == DVec::RANGE && pv.step == 1 && !pv.valid) {
                    DVec pos = partition_positions(entries(sd), pv.from, pv.n, order_only[(size_t)n.id] != 0);       // every entry holds a value: a permutation of 0 .. m-1
                    DVec r = make_sparse(sd.sel, pos.data);
                    r.perm = true;
                    r.iota = pos.iota;
                    r.order = pos.order; r.sorted_keys = pos.sorted_keys; r.sorted_keys_src = pos.sorted_keys_src;
                    return r;
                }
            }
            DVec data = densify(V(n.a));
            const DVec &piv = V(n.b);
            if (!(piv.kind == DVec::RANGE && piv.step == 1 && !piv.valid))
                throw Error(VDL_ERR_UNSUPPORTED, "Partition (Id " + std::to_string(n.id) +
                                                     "): pivots must be a RangeC with step 1 (what mplan2vdl emits, Vlite.hs:1088-1091)");
            return partition_positions(data, piv.from, piv.n, order_only[(size_t)n.id] != 0);
        }
        case Op::Semisort: {
            // gather mask that sorts the non-EPS values (stable): positions over [min, max] of the data, inverted
            DVec d = densify(V(n.a));
            o.kind = DVec::DENSE; o.n = d.n;
            o.data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(d.n, 1));
            o.valid = dev_alloc(c, sizeof(uint64_t) * (size_t)std::max<int64_t>(nwords(d.n), 1));
            HIP_CHECK(launch_fill_words((uint64_t *)o.valid->p, 0, nwords(d.n), s));
            if (d.n == 0) return o;
            BufP scratch = dev_alloc(c, sizeof(int64_t) * 3 * (size_t)fold_scratch_blocks());
            BufP mm = dev_alloc(c, 6 * sizeof(int64_t));
            HIP_CHECK(launch_fold_global(1, src_of(d), vp(d), nullptr, d.n, (int64_t *)scratch->p, (int64_t *)mm->p, s));
            HIP_CHECK(launch_fold_global(2, src_of(d), vp(d), nullptr, d.n, (int64_t *)scratch->p, (int64_t *)mm->p + 3, s));
            int64_t h[6];
            fetch_words(mm->p, 6, h);
            if (h[2] == 0) return o;                                      // nothing but EPS
            const uint64_t span = (uint64_t)h[3] - (uint64_t)h[0];
            if (span >= ((uint64_t)1 << 62))
                throw Error(VDL_ERR_UNSUPPORTED, "Semisort (Id " + std::to_string(n.id) + "): value range wider than 2^62");
            DVec pos = partition_positions(d, h[0], (int64_t)span + 1);
            Src iota; iota.kind = SRC_RANGE; iota.from = 0; iota.step = 1;
            HIP_CHECK(launch_scatter(iota, vp(d), src_of(pos), vp(pos), d.n, d.n, (int64_t *)o.data->p, (uint64_t *)o.valid->p, s));
            return o;
        }
        case Op::Materialize:
            materialize(n, V(n.a));
            return V(n.a);
        case Op::Cross: {
            const int64_t m = V(n.a).n, k = V(n.b).n;
            if (k > 0 && m > ((int64_t)1 << 40) / k)
                throw Error(VDL_ERR_NOMEM, "CrossProduct (Id " + std::to_string(n.id) + "): " + std::to_string(m) + " x " + std::to_string(k) + " slots");
            o.kind = DVec::DENSE; o.n = m * k;
            o.data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(o.n, 1));
            HIP_CHECK(launch_cross(o.n, k, n.bin, (int64_t *)o.data->p, s));
            return o;
        }
        case Op::Like: {
            if (sparse_on && V(n.a).kind == DVec::SPARSE) {
                const DVec &sd = V(n.a);
                DVec heap = densify(V(n.b));
                LikePattern pat{};
                pat.len = (int)n.pattern.size();
                memcpy(pat.p, n.pattern.data(), n.pattern.size());
                BufP data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(sd.sel->m, 1));
                HIP_CHECK(launch_like(i64_src(sd.data), nullptr, sd.sel->m, src_of(heap), vp(heap), heap.n, pat, (int64_t *)data->p, s));
                return make_sparse(sd.sel, data);
            }
            DVec d = densify(V(n.a)), heap = densify(V(n.b));
            LikePattern pat{};
            pat.len = (int)n.pattern.size();
            memcpy(pat.p, n.pattern.data(), n.pattern.size());
            o.kind = DVec::DENSE; o.n = d.n; o.valid = d.valid;
            o.data = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(d.n, 1));
            HIP_CHECK(launch_like(src_of(d), vp(d), d.n, src_of(heap), vp(heap), heap.n, pat, (int64_t *)o.data->p, s));
            return o;
        }
        }
        return o;
    }

    // vdl_plan_set_trace: host copy of what the statement's vector MEANS (n slots, value + holds-a-value), whatever form
    // it is stored in.  Reads the stored buffers only: no cache of the executor is touched (no bitmap_of / sel_for /
    // densify), so a traced run takes the same decisions as an untraced one.
    static const char *form_name(const DVec &r) {
        static const char *const kn[] = {"none", "dense", "column", "range", "onehot", "ohconst", "sparse", "expr", "lazy"};
        static_assert(sizeof kn / sizeof kn[0] == DVec::LAZYG + 1, "one name per vector form");
        return kn[r.kind];
    }
    template <typename T> std::vector<T> fetch(const void *dev, size_t count) {
        std::vector<T> h(count);
        const size_t words = (sizeof(T) * count + 7) / 8;
        int64_t *pin = count ? c->pinned((int64_t)words) : nullptr;          // (traces of small vectors: pinned; long ones: pageable)
        if (count) HIP_CHECK(hipMemcpyAsync(pin ? (void *)pin : (void *)h.data(), dev, sizeof(T) * count, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (pin) std::memcpy(h.data(), pin, sizeof(T) * count);
        return h;
    }
    void snapshot(const Node &n, const DVec &v) {
        Traced t;
        t.node = n.id; t.form = form_name(v); t.n = v.n;
        const bool stored = v.kind == DVec::DENSE || v.kind == DVec::COLUMN || v.kind == DVec::RANGE;
        if (v.n < 0 || v.n > kTraceMaxSlots || v.kind == DVec::NONE || v.kind == DVec::EXPR || v.kind == DVec::LAZYG) { p->traced.push_back(std::move(t)); return; }
        const size_t ns = (size_t)v.n;
        t.have = true;
        t.vals.assign(ns, 0); t.ok.assign(ns, 0);
        if (stored) {
            if (ns) {
                BufP tmp = dev_alloc(c, sizeof(int64_t) * ns);
                Src zero; zero.kind = SRC_RANGE; zero.from = 0; zero.step = 0;
                HIP_CHECK(launch_binary(B_ADD, src_of(v), zero, (int64_t *)tmp->p, v.n, s));
                t.vals = fetch<int64_t>(tmp->p, ns);
            }
            if (v.valid) {
                const std::vector<uint64_t> w = fetch<uint64_t>(v.valid->p, (size_t)nwords(v.n));
                for (size_t i = 0; i < ns; i++) t.ok[i] = (uint8_t)((w[i >> 6] >> (i & 63)) & 1u);
            } else t.ok.assign(ns, 1);
            for (size_t i = 0; i < ns; i++) if (!t.ok[i]) t.vals[i] = 0;
        } else if (v.kind == DVec::SPARSE) {
            const size_t m = (size_t)v.sel->m;
            const std::vector<int64_t> data = fetch<int64_t>(v.data->p, m);
            std::vector<int64_t> idx(m);
            if (v.sel->idx) idx = fetch<int64_t>(v.sel->idx->p, m); else for (size_t k = 0; k < m; k++) idx[k] = (int64_t)k;
            std::vector<uint64_t> held;
            if (v.valid) held = fetch<uint64_t>(v.valid->p, (size_t)nwords((int64_t)m));
            for (size_t k = 0; k < m; k++) {
                if (v.valid && !((held[k >> 6] >> (k & 63)) & 1u)) continue;
                if (idx[k] < 0 || idx[k] >= v.n || t.ok[(size_t)idx[k]]) { t.form = "sparse(BROKEN SELECTION)"; continue; }   // slot ids must be distinct and in range
                t.vals[(size_t)idx[k]] = data[k]; t.ok[(size_t)idx[k]] = 1;
            }
        } else {                                                  // ONEHOT / OHCONST: {value, slot, count} of a global fold
            const std::vector<int64_t> rec = fetch<int64_t>(v.data->p, 3);
            if (rec[2] > 0 && rec[1] >= 0 && rec[1] < v.n) {
                t.vals[(size_t)rec[1]] = v.kind == DVec::OHCONST ? v.from : rec[0];
                t.ok[(size_t)rec[1]] = 1;
            }
        }
        p->traced.push_back(std::move(t));
    }

    void run() { run_nodes(p->prog.outputs, nullptr); }

    // Executes the statements `targets` depend on.  `overrides` supplies ready-made vectors for some
    // statements (their own operands are then not evaluated): used by the sharded Partition exchange.
    void run_nodes(const std::vector<int> &targets, const std::map<int, DVec> *overrides) {
        const Program &P = p->prog;
        std::vector<char> needed(P.nodes.size(), 0);
        done.assign(P.nodes.size(), 0);
        cur_over = overrides;
        for (int id : targets) needed[(size_t)id] = 1;
        for (auto it = P.order.rbegin(); it != P.order.rend(); ++it) {
            const Node &n = P.at(*it);
            if (!needed[(size_t)n.id]) continue;
            if (overrides && overrides->count(n.id)) continue;
            if (column_filter(n)) continue;
            for (int opnd : {n.a, n.b, n.c}) if (opnd > 0) needed[(size_t)opnd] = 1;
        }
        for (size_t k = 0; k < P.order.size(); k++) {
            const Node &n = P.at(P.order[k]);
            if (!needed[(size_t)n.id] || (overrides && overrides->count(n.id)) || column_filter(n)) continue;
            for (int opnd : {n.a, n.b, n.c}) if (opnd > 0) last_use[(size_t)opnd] = (int)k;
        }
        for (int id : targets) last_use[(size_t)id] = 1 << 30;      // targets stay alive for the caller
        n_uses.assign(P.nodes.size(), 0);
        read_by_binary_only.assign(P.nodes.size(), 1);
        value_uses.assign(P.nodes.size(), 0);
        value_read_by_binary_only.assign(P.nodes.size(), 1);
        std::vector<std::vector<std::pair<int, int>>> readers(P.nodes.size());      // (reader id, operand slot)
        for (size_t k = 0; k < P.order.size(); k++) {
            const Node &n = P.at(P.order[k]);
            if (!needed[(size_t)n.id] || (overrides && overrides->count(n.id)) || column_filter(n)) continue;
            int slot = 0;
            for (int opnd : {n.a, n.b, n.c}) {
                if (opnd > 0) {
                    n_uses[(size_t)opnd]++;
                    if (n.op != Op::Binary) read_by_binary_only[(size_t)opnd] = 0;
                    if (n.op != Op::RangeV) {                      // (RangeV reads the shape only)
                        value_uses[(size_t)opnd]++;
                        if (n.op != Op::Binary) value_read_by_binary_only[(size_t)opnd] = 0;
                    }
                    readers[(size_t)opnd].push_back({n.id, slot});
                }
                slot++;
            }
            if (n.op == Op::Binary && n.a == n.b && n.a > 0) { n_uses[(size_t)n.a]--; value_uses[(size_t)n.a]--; }       // x op x: both operands are one reader
        }
        for (int id : targets) { n_uses[(size_t)id] += 2; value_uses[(size_t)id] += 2; }
        lazy_gather_ok.assign(P.nodes.size(), 0);
        for (int id : P.order) {
            const Node &g = P.at(id);
            if (g.op != Op::Gather || !needed[(size_t)id] || n_uses[(size_t)id] != (int)readers[(size_t)id].size()) continue;   // targets excluded
            const auto &rd = readers[(size_t)id];
            if (rd.size() == 1 && P.at(rd[0].first).op == Op::Gather && rd[0].second == 0 && P.at(rd[0].first).b != id) lazy_gather_ok[(size_t)id] = 1;
            if (rd.size() == 2) {
                int rv = -1, fs = -1;
                for (const auto &r : rd) {
                    const Node &x = P.at(r.first);
                    if (x.op == Op::RangeV && x.imm1 != 0 && r.second == 0) rv = r.first;
                    if (x.op == Op::FoldSelect && r.second == 1 && x.b == id) fs = r.first;
                }
                if (rv >= 0 && fs >= 0 && P.at(fs).a == rv && readers[(size_t)rv].size() == 1 && n_uses[(size_t)rv] == 1) lazy_gather_ok[(size_t)id] = 1;
            }
        }
        lazy_pred_ok.assign(P.nodes.size(), 0);
        for (int id : P.order) {
            const Node &b = P.at(id);
            if (b.op != Op::Binary || !needed[(size_t)id] || n_uses[(size_t)id] != (int)readers[(size_t)id].size()) continue;   // targets excluded
            const auto &rd = readers[(size_t)id];
            if (rd.size() != 2) continue;
            int rv = -1, fs = -1;
            for (const auto &r : rd) {
                const Node &x = P.at(r.first);
                if (x.op == Op::RangeV && x.imm1 != 0 && r.second == 0) rv = r.first;
                if (x.op == Op::FoldSelect && r.second == 1 && x.b == id) fs = r.first;
            }
            if (rv >= 0 && fs >= 0 && P.at(fs).a == rv && readers[(size_t)rv].size() == 1 && n_uses[(size_t)rv] == 1) lazy_pred_ok[(size_t)id] = 1;
        }
        order_only.assign(P.nodes.size(), 0);
        for (int id : P.order) {
            const Node &pn = P.at(id);
            if (pn.op != Op::Partition || !needed[(size_t)id] || n_uses[(size_t)id] != (int)readers[(size_t)id].size() || readers[(size_t)id].empty()) continue;   // targets excluded
            bool all = !p->tracing;                             // (a traced run shows the positions)
            for (const auto &r : readers[(size_t)id]) all = all && P.at(r.first).op == Op::Scatter && r.second == 2 && P.at(r.first).a != id && P.at(r.first).b != id;
            order_only[(size_t)id] = all ? 1 : 0;
        }
        needed_now = needed;
        p->outs.clear();
        p->timings.clear();
        if (p->profiling && !p->stmt_ev[1]) { HIP_CHECK(hipEventCreate(&p->stmt_ev[0])); HIP_CHECK(hipEventCreate(&p->stmt_ev[1])); }
        const hipEvent_t e0 = p->stmt_ev[0], e1 = p->stmt_ev[1];      // owned by the plan: an error exit leaks nothing
        p->traced.clear();
        for (size_t k = 0; k < P.order.size(); k++) {
            const Node &n = P.at(P.order[k]);
            if (!needed[(size_t)n.id]) continue;
            if (overrides && overrides->count(n.id)) { vec[(size_t)n.id] = overrides->at(n.id); continue; }
            if (done[(size_t)n.id]) {                              // ran ahead of its turn (ensure): only its operands' lifetimes end here
                for (int opnd : {n.a, n.b, n.c})
                    if (opnd > 0 && last_use[(size_t)opnd] == (int)k) vec[(size_t)opnd] = DVec{};
                continue;
            }
            if (p->profiling) HIP_CHECK(hipEventRecord(e0, s));
            vec[(size_t)n.id] = exec(n);
            if (p->tracing) snapshot(n, vec[(size_t)n.id]);
            if (trace_forms) {
                const DVec &r = vec[(size_t)n.id];
                std::fprintf(stderr, "  Id %-4d %-18s -> %-7s n=%lld", n.id, op_name(n.op, n.bin), form_name(r), (long long)r.n);
                if (r.kind == DVec::SPARSE) std::fprintf(stderr, " m=%lld%s%s", (long long)r.sel->m, r.sel->idx ? "" : " (prefix)", r.perm ? " perm" : "");
                std::fprintf(stderr, "  scatters so far %d\n", densified);
            }
            if (p->profiling) {
                HIP_CHECK(hipEventRecord(e1, s));
                HIP_CHECK(hipEventSynchronize(e1));
                float ms = 0;
                HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
                p->timings.push_back({"timeInMicrosecondsForStatement" + std::to_string(n.id) + "_" + op_name(n.op, n.bin), (double)ms * 1e3});
            }
            for (int opnd : {n.a, n.b, n.c})
                if (opnd > 0 && last_use[(size_t)opnd] == (int)k) vec[(size_t)opnd] = DVec{};
        }
        HIP_CHECK(hipStreamSynchronize(s));
        finish_copies();
    }
};

}  // namespace eng
}  // namespace vdl

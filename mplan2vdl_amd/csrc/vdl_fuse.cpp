// vdl_fuse.cpp -- symbolic analysis of a VDL program and scan fusion (see vdl_fuse.h).
//
// Every vector is classified as
//   ROW  : one value per row of a base table, value = expression over that table's loaded
//          columns / constants / row ids, EPS exactly where a *selection* rejects the row;
//   FOLD : result of a global fold (single run) -- a scalar living in one slot;
//   NONE : anything else (the program then runs operator by operator).
// Selections come from FoldSelect(RangeV 0 1 d, d) (/root/reference/src/Vlite.hs:725-727),
// are propagated by Gather (Vlite.hs:729) and -- in this repo's vector model -- by RangeV.
#include "vdl_fuse.h"
#include "vdl.h"

#include <algorithm>
#include <cstdlib>
#include <sstream>

namespace vdl {
namespace {

struct Row;
using RowP = std::shared_ptr<const Row>;
struct Row {
    enum K { COL, CONST, IOTA, BIN } k = CONST;
    std::string col;
    int64_t c0 = 0, c1 = 0;  // CONST: value; IOTA: from, step
    int bin = -1;
    RowP l, r;
};

RowP mk_col(const std::string &c) { auto p = std::make_shared<Row>(); p->k = Row::COL; p->col = c; return p; }
RowP mk_const(int64_t v) { auto p = std::make_shared<Row>(); p->k = Row::CONST; p->c0 = v; return p; }
RowP mk_iota(int64_t f, int64_t s) { auto p = std::make_shared<Row>(); p->k = Row::IOTA; p->c0 = f; p->c1 = s; return p; }
RowP mk_bin(int op, RowP l, RowP r) {
    if (l->k == Row::CONST && r->k == Row::CONST) return mk_const(apply_bin(op, l->c0, r->c0));
    auto p = std::make_shared<Row>(); p->k = Row::BIN; p->bin = op; p->l = std::move(l); p->r = std::move(r); return p;
}


// SUM of a constant k over the selected rows = k * (number of selected rows), and every scan already carries
// that count as word 0 of its partials (count(*) lowers to FoldSum of ones, Vlite.hs:1044-1046): no aggregate
// slot, no per-row work.  agg == -1 addresses the count word (eval_scalar is handed a pointer to word 1).
static ScalarP count_times(int64_t k) {
    auto cnt = std::make_shared<Scalar>(); cnt->k = Scalar::AGG; cnt->agg = -1;
    if (k == 1) return cnt;
    auto kk = std::make_shared<Scalar>(); kk->k = Scalar::CONST; kk->c = k;
    auto m = std::make_shared<Scalar>(); m->k = Scalar::BIN; m->bin = B_MUL; m->l = cnt; m->r = kk;
    return m;
}

struct Sym {
    //  RANGEC : RangeC from count step (pivots)
    //  PART   : Partition(ROW key, RangeC min cnt 1) -- positions that sort the selected rows by key
    //  SORTED : Scatter(ROW x, _, PART) -- x in key order
    //  GFOLD  : Fold(SORTED key, SORTED x) -- one value per distinct key = per group
    enum Kind { NONE, ROW, FOLD, RANGEC, PART, SORTED, GFOLD } kind = NONE;
    std::string table;
    int sel = 0;       // ROW/SORTED/PART: selection index (0 = every row); FOLD: selection of the control vector
    RowP e;            // ROW, SORTED (source expression), PART (key expression)
    int scan = -1;     // FOLD: scan index; GFOLD: group-scan index
    ScalarP sc;        // FOLD, GFOLD
    int part = -1;     // PART/SORTED: node id of the Partition
    int64_t r_from = 0, r_count = 0, r_step = 0;   // RANGEC; PART: pivots
};

bool row_equal(const RowP &a, const RowP &b) {
    if (a == b) return true;
    if (!a || !b || a->k != b->k) return false;
    switch (a->k) {
    case Row::COL: return a->col == b->col;
    case Row::CONST: return a->c0 == b->c0;
    case Row::IOTA: return a->c0 == b->c0 && a->c1 == b->c1;
    default: return a->bin == b->bin && row_equal(a->l, b->l) && row_equal(a->r, b->r);
    }
}

struct Selection { std::string table; RowP pred; };   // pred == nullptr: all rows

// ---- closed-interval sets over int64 ---------------------------------------------
using Iv = std::pair<int64_t, int64_t>;
using IvSet = std::vector<Iv>;                        // sorted, disjoint, non-adjacent

IvSet iv_norm(IvSet s) {
    std::sort(s.begin(), s.end());
    IvSet o;
    for (auto &x : s) {
        if (x.first > x.second) continue;
        if (!o.empty() && (o.back().second == INT64_MAX || x.first <= o.back().second + 1))
            o.back().second = std::max(o.back().second, x.second);
        else o.push_back(x);
    }
    return o;
}
IvSet iv_and(const IvSet &a, const IvSet &b) {
    IvSet o;
    for (auto &x : a) for (auto &y : b) {
        int64_t lo = std::max(x.first, y.first), hi = std::min(x.second, y.second);
        if (lo <= hi) o.push_back({lo, hi});
    }
    return iv_norm(o);
}
IvSet iv_or(IvSet a, const IvSet &b) { a.insert(a.end(), b.begin(), b.end()); return iv_norm(a); }

// conjunction over columns of (column in IvSet); `never` = constant false
struct Clause { std::map<std::string, IvSet> cols; bool never = false; };

bool leaf_cmp(const Row &p, std::string &col, IvSet &set) {
    if (p.k != Row::BIN) return false;
    const Row &l = *p.l, &r = *p.r;
    if (p.bin == B_GT) {
        if (l.k == Row::COL && r.k == Row::CONST) {       // col > k
            col = l.col; set = r.c0 == INT64_MAX ? IvSet{} : IvSet{{r.c0 + 1, INT64_MAX}}; return true;
        }
        if (l.k == Row::CONST && r.k == Row::COL) {       // k > col
            col = r.col; set = l.c0 == INT64_MIN ? IvSet{} : IvSet{{INT64_MIN, l.c0 - 1}}; return true;
        }
    } else if (p.bin == B_EQ) {
        if (l.k == Row::COL && r.k == Row::CONST) { col = l.col; set = {{r.c0, r.c0}}; return true; }
        if (l.k == Row::CONST && r.k == Row::COL) { col = r.col; set = {{l.c0, l.c0}}; return true; }
    }
    return false;
}

// predicate "p != 0" -> Clause; false when p has a shape the scan kernel cannot express
bool to_clause(const RowP &p, Clause &out) {
    if (!p) return true;
    if (p->k == Row::CONST) { if (p->c0 == 0) out.never = true; return true; }
    if (p->k == Row::COL) {
        IvSet nz{{INT64_MIN, -1}, {1, INT64_MAX}};
        auto it = out.cols.find(p->col);
        if (it == out.cols.end()) out.cols[p->col] = nz; else it->second = iv_and(it->second, nz);
        return true;
    }
    if (p->k != Row::BIN) return false;
    std::string col; IvSet set;
    if (leaf_cmp(*p, col, set)) {
        auto it = out.cols.find(col);
        if (it == out.cols.end()) out.cols[col] = iv_norm(set); else it->second = iv_and(it->second, set);
        return true;
    }
    if (p->bin == B_LAND) return to_clause(p->l, out) && to_clause(p->r, out);
    if (p->bin == B_LOR) {
        // only a disjunction of conditions on ONE column stays a per-column filter
        // (e.g. `<=` printed as LogicalOr(Greater, Equals), /root/reference/src/Vdl.hs:143-144)
        Clause a, b;
        if (!to_clause(p->l, a) || !to_clause(p->r, b)) return false;
        if (a.never && b.never) { out.never = true; return true; }
        if (a.never) { for (auto &kv : b.cols) { auto it = out.cols.find(kv.first); if (it == out.cols.end()) out.cols[kv.first] = kv.second; else it->second = iv_and(it->second, kv.second); } return true; }
        if (b.never) { for (auto &kv : a.cols) { auto it = out.cols.find(kv.first); if (it == out.cols.end()) out.cols[kv.first] = kv.second; else it->second = iv_and(it->second, kv.second); } return true; }
        if (a.cols.empty() || b.cols.empty()) return true;   // one side is constant true
        if (a.cols.size() != 1 || b.cols.size() != 1 || a.cols.begin()->first != b.cols.begin()->first) return false;
        IvSet u = iv_or(a.cols.begin()->second, b.cols.begin()->second);
        const std::string &c = a.cols.begin()->first;
        auto it = out.cols.find(c);
        if (it == out.cols.end()) out.cols[c] = u; else it->second = iv_and(it->second, u);
        return true;
    }
    return false;
}

// ---- aggregate data: product of affine single-column factors ----------------------
struct Affine { bool has_col = false; std::string col; int64_t a = 0, s = 0; };   // a + s*col

bool to_affine(const RowP &e, Affine &out) {
    switch (e->k) {
    case Row::CONST: out = Affine{false, "", e->c0, 0}; return true;
    case Row::COL: out = Affine{true, e->col, 0, 1}; return true;
    case Row::IOTA: return false;
    case Row::BIN: break;
    }
    Affine x, y;
    if (e->bin == B_ADD || e->bin == B_SUB) {
        if (!to_affine(e->l, x) || !to_affine(e->r, y)) return false;
        if (x.has_col && y.has_col && x.col != y.col) return false;
        int64_t sign = e->bin == B_ADD ? 1 : -1;
        out.has_col = x.has_col || y.has_col;
        out.col = x.has_col ? x.col : y.col;
        out.a = apply_bin(B_ADD, x.a, apply_bin(B_MUL, sign, y.a));
        out.s = apply_bin(B_ADD, x.s, apply_bin(B_MUL, sign, y.s));
        return true;
    }
    if (e->bin == B_MUL) {
        if (!to_affine(e->l, x) || !to_affine(e->r, y)) return false;
        if (x.has_col && y.has_col) return false;          // quadratic: handled as two factors
        const Affine &k = x.has_col ? y : x, &v = x.has_col ? x : y;
        out.has_col = v.has_col; out.col = v.col;
        out.a = apply_bin(B_MUL, v.a, k.a);
        out.s = apply_bin(B_MUL, v.s, k.a);
        return true;
    }
    return false;
}

bool to_product(const RowP &e, std::vector<Affine> &fac) {
    Affine f;
    if (to_affine(e, f)) { fac.push_back(f); return true; }
    if (e->k == Row::BIN && e->bin == B_MUL) return to_product(e->l, fac) && to_product(e->r, fac);
    return false;
}

struct Builder {
    const Program &P;
    std::vector<Sym> sym;
    std::vector<Selection> sels;                               // 1-based via sels[i-1]
    std::map<std::pair<int, int>, int> conj_memo;
    struct PendingScan { std::string table; int sel; std::vector<RowP> data; std::vector<int> kind; };
    std::vector<PendingScan> scans;
    struct PendingGroup { int part; std::string table; int sel; RowP key; int64_t pmin, pcount; std::vector<RowP> data; std::vector<int> kind; };
    std::vector<PendingGroup> groups;
    std::string why;
    std::map<int, FilterSpec> filters;

    explicit Builder(const Program &p) : P(p), sym(p.nodes.size()) {}

    int new_sel(const std::string &table, RowP pred) { sels.push_back({table, std::move(pred)}); return (int)sels.size(); }
    RowP pred_of(int s) const { return s ? sels[(size_t)s - 1].pred : nullptr; }

    int combine_sel(const std::string &table, int a, int b) {
        if (a == b || b == 0) return a;
        if (a == 0) return b;
        auto key = std::make_pair(std::min(a, b), std::max(a, b));
        auto it = conj_memo.find(key);
        if (it != conj_memo.end()) return it->second;
        int s = new_sel(table, mk_bin(B_LAND, pred_of(a), pred_of(b)));
        conj_memo[key] = s;
        return s;
    }

    int scan_for(const std::string &table, int sel) {
        for (size_t i = 0; i < scans.size(); i++) if (scans[i].table == table && scans[i].sel == sel) return (int)i;
        scans.push_back({table, sel, {}, {}});
        return (int)scans.size() - 1;
    }

    Sym visit(const Node &n) {
        Sym out;
        auto S = [&](int id) -> const Sym & { return sym[(size_t)id]; };
        switch (n.op) {
        case Op::Load: {
            out.kind = Sym::ROW;
            size_t dot = n.column.find('.');
            out.table = dot == std::string::npos ? n.column : n.column.substr(0, dot);
            out.e = mk_col(n.column);
            return out;
        }
        case Op::Project: case Op::Shuffle: case Op::Materialize:
            return S(n.a);
        case Op::RangeC:
            out.kind = Sym::RANGEC; out.r_from = n.imm0; out.r_count = n.imm1; out.r_step = n.imm2;
            return out;
        case Op::Partition: {
            // group-by scatter mask, /root/reference/src/Vlite.hs:1082-1098
            const Sym &d = S(n.a), &pv = S(n.b);
            if (d.kind != Sym::ROW || pv.kind != Sym::RANGEC || pv.r_step != 1 || pv.r_count <= 0) return out;
            out.kind = Sym::PART; out.table = d.table; out.sel = d.sel; out.e = d.e; out.part = n.id;
            out.r_from = pv.r_from; out.r_count = pv.r_count;
            return out;
        }
        case Op::Scatter: {
            // sorted keys / sorted aggregate inputs, Vlite.hs:1058-1059
            const Sym &src = S(n.a), &fold = S(n.b), &pos = S(n.c);
            if (src.kind != Sym::ROW || pos.kind != Sym::PART || fold.kind != Sym::ROW) return out;
            if (src.table != pos.table || fold.table != pos.table) return out;
            out.kind = Sym::SORTED; out.table = src.table; out.e = src.e; out.part = pos.part;
            out.sel = combine_sel(src.table, src.sel, pos.sel);
            return out;
        }
        case Op::RangeV: {
            const Sym &r = S(n.a);
            if (r.kind == Sym::ROW) {
                out.kind = Sym::ROW; out.table = r.table; out.sel = r.sel;
                out.e = n.imm1 == 0 ? mk_const(n.imm0) : mk_iota(n.imm0, n.imm1);
            } else if ((r.kind == Sym::FOLD || r.kind == Sym::GFOLD) && n.imm1 == 0) {
                out = r;
                auto s = std::make_shared<Scalar>(); s->k = Scalar::CONST; s->c = n.imm0; out.sc = s;
            }
            return out;
        }
        case Op::Binary: {
            const Sym &a = S(n.a), &b = S(n.b);
            if (a.kind == Sym::ROW && b.kind == Sym::ROW && a.table == b.table) {
                out.kind = Sym::ROW; out.table = a.table;
                out.sel = combine_sel(a.table, a.sel, b.sel);
                out.e = mk_bin(n.bin, a.e, b.e);
            } else if ((a.kind == Sym::FOLD || a.kind == Sym::GFOLD) && a.kind == b.kind && a.scan == b.scan && a.sel == b.sel) {
                out = a;
                auto s = std::make_shared<Scalar>(); s->k = Scalar::BIN; s->bin = n.bin; s->l = a.sc; s->r = b.sc; out.sc = s;
            }
            return out;
        }
        case Op::FoldSelect: {
            const Sym &ctl = S(n.a), &d = S(n.b);
            if (ctl.kind != Sym::ROW || d.kind != Sym::ROW || ctl.table != d.table) return out;
            if (ctl.e->k != Row::IOTA || ctl.e->c1 == 0) return out;      // runs of length one only
            int base = combine_sel(d.table, ctl.sel, d.sel);
            RowP pred = base ? mk_bin(B_LAND, pred_of(base), d.e) : d.e;
            if (!base) {
                Clause cl;
                if (to_clause(d.e, cl) && (cl.never || (!cl.cols.empty() && (int)cl.cols.size() <= kMaxFilterCols))) {
                    FilterSpec fs;
                    fs.table = d.table; fs.never = cl.never;
                    bool fits = true;
                    for (auto &kv : cl.cols) {
                        FilterColumn fc;
                        fc.name = kv.first;
                        if ((int)kv.second.size() > kMaxFilterIvs) { fits = false; break; }
                        if (kv.second.empty()) fs.never = true;
                        for (auto &iv : kv.second) { fc.lo[fc.n] = iv.first; fc.hi[fc.n] = iv.second; fc.n++; }
                        fs.cols.push_back(fc);
                    }
                    if (fits && !fs.cols.empty()) filters[n.id] = fs;
                }
            }
            out.kind = Sym::ROW; out.table = d.table; out.e = mk_iota(0, 1);
            out.sel = new_sel(d.table, pred);
            return out;
        }
        case Op::Gather: {
            const Sym &src = S(n.a), &pos = S(n.b);
            if (src.kind != Sym::ROW || pos.kind != Sym::ROW || src.table != pos.table) return out;
            if (pos.e->k != Row::IOTA || pos.e->c0 != 0 || pos.e->c1 != 1) return out;   // identity positions with holes
            out.kind = Sym::ROW; out.table = src.table; out.e = src.e;
            out.sel = combine_sel(src.table, src.sel, pos.sel);
            return out;
        }
        case Op::FoldSum: case Op::FoldMin: case Op::FoldMax: case Op::FoldCount: case Op::FoldChoose: {
            const Sym &ctl = S(n.a), &d = S(n.b);
            if (ctl.kind == Sym::SORTED && d.kind == Sym::SORTED && ctl.part == d.part) {
                // grouped aggregate: fold of data scattered into key order over the sorted key
                // (Vlite.hs:1056-1060).  Exact only if control and data cover the same rows.
                const Sym &pt = sym[(size_t)ctl.part];
                if (!row_equal(ctl.e, pt.e) || ctl.sel != pt.sel || d.sel != pt.sel) return out;
                int g = -1;
                for (size_t i = 0; i < groups.size(); i++) if (groups[i].part == ctl.part) g = (int)i;
                if (g < 0) { groups.push_back({ctl.part, pt.table, pt.sel, pt.e, pt.r_from, pt.r_count, {}, {}}); g = (int)groups.size() - 1; }
                PendingGroup &pg = groups[(size_t)g];
                int kind = n.op == Op::FoldMin ? AGG_MIN : n.op == Op::FoldMax ? AGG_MAX : n.op == Op::FoldChoose ? AGG_FIRST : AGG_SUM;
                if (kind == AGG_FIRST && d.e->k != Row::COL) return out;
                RowP data = n.op == Op::FoldCount ? mk_const(1) : d.e;
                if (kind == AGG_SUM && data->k == Row::CONST) {
                    out.kind = Sym::GFOLD; out.table = pt.table; out.sel = pt.sel; out.scan = g; out.sc = count_times(data->c0);
                    return out;
                }
                int idx = -1;      // the emitter repeats folds (CSE keyed on metadata, Vdl.hs:302,314-320): share them
                for (size_t i = 0; i < pg.data.size(); i++) if (pg.kind[i] == kind && row_equal(pg.data[i], data)) idx = (int)i;
                if (idx < 0) { pg.data.push_back(data); pg.kind.push_back(kind); idx = (int)pg.data.size() - 1; }
                out.kind = Sym::GFOLD; out.table = pt.table; out.sel = pt.sel; out.scan = g;
                auto s = std::make_shared<Scalar>(); s->k = Scalar::AGG; s->agg = idx; out.sc = s;
                return out;
            }
            if (n.op == Op::FoldChoose) return out;
            if (ctl.kind != Sym::ROW || d.kind != Sym::ROW || ctl.table != d.table) return out;
            if (ctl.e->k != Row::CONST) return out;                       // single run = global fold
            int eff = combine_sel(d.table, ctl.sel, d.sel);
            int sc = scan_for(d.table, eff);
            PendingScan &ps = scans[(size_t)sc];
            int kind = n.op == Op::FoldMin ? AGG_MIN : n.op == Op::FoldMax ? AGG_MAX : AGG_SUM;
            RowP data = n.op == Op::FoldCount ? mk_const(1) : d.e;
            if (kind == AGG_SUM && data->k == Row::CONST) {
                out.kind = Sym::FOLD; out.table = d.table; out.sel = ctl.sel; out.scan = sc; out.sc = count_times(data->c0);
                return out;
            }
            int idx = -1;
            for (size_t i = 0; i < ps.data.size(); i++) if (ps.kind[i] == kind && row_equal(ps.data[i], data)) idx = (int)i;
            if (idx < 0) { ps.data.push_back(data); ps.kind.push_back(kind); idx = (int)ps.data.size() - 1; }
            out.kind = Sym::FOLD; out.table = d.table; out.sel = ctl.sel; out.scan = sc;
            auto s = std::make_shared<Scalar>(); s->k = Scalar::AGG; s->agg = idx; out.sc = s;
            return out;
        }
        default:
            return out;
        }
    }
};

void show_row(const RowP &e, std::ostringstream &o) {
    switch (e->k) {
    case Row::COL: o << e->col; break;
    case Row::CONST: o << e->c0; break;
    case Row::IOTA: o << "iota(" << e->c0 << "," << e->c1 << ")"; break;
    case Row::BIN: o << kBinNames[e->bin] << "("; show_row(e->l, o); o << ","; show_row(e->r, o); o << ")"; break;
    }
}


// key expression -> two-accumulator program (vdl_fuse.h KeyStep); false if the tree is not
// left/right-deep with single-column leaves
bool single_column_chain(const RowP &e) {
    if (e->k == Row::COL) return true;
    if (e->k != Row::BIN) return false;
    if (e->r->k == Row::CONST) return single_column_chain(e->l);
    if (e->l->k == Row::CONST) return single_column_chain(e->r);
    return false;
}

template <typename ColIndex>
bool emit_key(const RowP &e, int target, std::vector<KeyStep> &prog, ColIndex &col_index) {
    KeyStep st;
    if (e->k == Row::COL) { st.kind = KeyStep::LOAD; st.target = target; st.col = col_index(e->col); prog.push_back(st); return true; }
    if (e->k != Row::BIN) return false;
    if (e->r->k == Row::CONST || e->l->k == Row::CONST) {
        const bool left = e->r->k != Row::CONST;
        if (!emit_key(left ? e->r : e->l, target, prog, col_index)) return false;
        if (e->bin == B_DIV || e->bin == B_MOD) return false;      // kept off the in-kernel key evaluator
        st.kind = KeyStep::OPK; st.target = target; st.bin = e->bin; st.const_left = left ? 1 : 0; st.k = left ? e->l->c0 : e->r->c0;
        prog.push_back(st);
        return true;
    }
    if (target != 0 || e->bin == B_DIV || e->bin == B_MOD) return false;
    if (single_column_chain(e->r)) {
        if (!emit_key(e->l, 0, prog, col_index) || !emit_key(e->r, 1, prog, col_index)) return false;
        st.kind = KeyStep::COMBINE; st.bin = e->bin; st.const_left = 0; prog.push_back(st);
        return true;
    }
    if (single_column_chain(e->l)) {
        if (!emit_key(e->r, 0, prog, col_index) || !emit_key(e->l, 1, prog, col_index)) return false;
        st.kind = KeyStep::COMBINE; st.bin = e->bin; st.const_left = 1; prog.push_back(st);
        return true;
    }
    return false;
}

// predicate + aggregate inputs -> ScanColumn filters and ScanAgg products; shared by scans and group scans
template <typename Plan>
bool lower_common(const RowP &pred, const std::vector<RowP> &data, const std::vector<int> &kind, Plan &sp, std::string &why) {
    Clause cl;
    if (!to_clause(pred, cl)) {
        std::ostringstream o; o << "predicate is not a conjunction of per-column ranges: ";
        show_row(pred, o);
        why = o.str();
        return false;
    }
    sp.never = cl.never;
    auto col_index = [&](const std::string &name) -> int {
        for (size_t i = 0; i < sp.cols.size(); i++) if (sp.cols[i].name == name) return (int)i;
        sp.cols.push_back(ScanColumn{name, INT64_MIN, INT64_MAX});
        return (int)sp.cols.size() - 1;
    };
    for (auto &kv : cl.cols) {
        if (kv.second.empty()) { sp.never = true; col_index(kv.first); continue; }
        if (kv.second.size() != 1) { why = "filter on " + kv.first + " is not a single range"; return false; }
        int c = col_index(kv.first);
        sp.cols[(size_t)c].lo = kv.second[0].first;
        sp.cols[(size_t)c].hi = kv.second[0].second;
    }
    for (size_t j = 0; j < data.size(); j++) {
        ScanAgg ag;
        ag.kind = kind[j];
        if (ag.kind == AGG_FIRST) {
            ag.fac.push_back(ScanFactor{col_index(data[j]->col), 0, 1});
            sp.aggs.push_back(ag);
            continue;
        }
        std::vector<Affine> fac;
        if (!to_product(data[j], fac)) {
            std::ostringstream o; o << "aggregate input is not a product of affine column factors: ";
            show_row(data[j], o);
            why = o.str();
            return false;
        }
        int64_t mult = 1;
        bool repeated = false;
        for (auto &f : fac) {
            if (!f.has_col || f.s == 0) { mult = apply_bin(B_MUL, mult, f.a); continue; }
            const int ci = col_index(f.col);
            for (auto &g : ag.fac) repeated |= g.col == ci;
            ag.fac.push_back(ScanFactor{ci, f.a, f.s});
        }
        if (repeated) { why = "a column appears twice in one aggregate product"; return false; }
        if (ag.fac.empty()) ag.constant = mult;
        else { ag.fac[0].a = apply_bin(B_MUL, ag.fac[0].a, mult); ag.fac[0].s = apply_bin(B_MUL, ag.fac[0].s, mult); }
        if ((int)ag.fac.size() > kMaxFactors) { why = "aggregate has more than 4 column factors"; return false; }
        sp.aggs.push_back(ag);
    }
    return true;
}

}  // namespace

int64_t eval_scalar(const Scalar &s, const int64_t *agg) {
    switch (s.k) {
    case Scalar::AGG: return agg[s.agg];
    case Scalar::CONST: return s.c;
    default: return apply_bin(s.bin, eval_scalar(*s.l, agg), eval_scalar(*s.r, agg));
    }
}

FusedPlan fuse_program(const Program &P) {
    FusedPlan F;
    Builder B(P);
    for (int id : P.order) B.sym[(size_t)id] = B.visit(P.at(id));
    F.filters = B.filters;
    if (P.outputs.empty()) { F.why_not = "program has no MaterializeCompact output"; return F; }
    for (int id : P.outputs) {
        const Sym &s = B.sym[(size_t)id];
        if (s.kind != Sym::FOLD && s.kind != Sym::GFOLD) {
            F.why_not = "output Id " + std::to_string(id) + " is neither a global fold nor a dense-domain grouped fold over filtered table columns";
            return F;
        }
    }
    // lower every pending scan to the kernel's clause / product form
    for (auto &ps : B.scans) {
        ScanPlan sp;
        sp.table = ps.table;
        if (!lower_common(B.pred_of(ps.sel), ps.data, ps.kind, sp, F.why_not)) return F;
        if (sp.cols.empty()) { F.why_not = "scan touches no column (row count unknown)"; return F; }
        if ((int)sp.cols.size() > kMaxScanCols) { F.why_not = "scan touches more than 8 columns"; return F; }
        if ((int)sp.aggs.size() > kMaxScanAggs) { F.why_not = "scan has more than 8 aggregates"; return F; }
        F.scans.push_back(sp);
    }
    for (auto &pg : B.groups) {
        GroupScanPlan gp;
        gp.table = pg.table; gp.pmin = pg.pmin; gp.pcount = pg.pcount;
        if (!lower_common(B.pred_of(pg.sel), pg.data, pg.kind, gp, F.why_not)) return F;
        auto col_index = [&](const std::string &name) -> int {
            for (size_t i = 0; i < gp.cols.size(); i++) if (gp.cols[i].name == name) return (int)i;
            gp.cols.push_back(ScanColumn{name, INT64_MIN, INT64_MAX});
            return (int)gp.cols.size() - 1;
        };
        if (!emit_key(pg.key, 0, gp.key, col_index)) {
            std::ostringstream o; o << "group key is not a chain of single-column terms: ";
            show_row(pg.key, o);
            F.why_not = o.str();
            return F;
        }
        if ((int)gp.key.size() > kMaxKeySteps) { F.why_not = "group key program too long"; return F; }
        if ((int)gp.cols.size() > kMaxScanCols) { F.why_not = "grouped scan touches more than 8 columns"; return F; }
        if ((int)gp.aggs.size() > 2 * kMaxScanAggs) { F.why_not = "grouped scan has more than 16 aggregates"; return F; }
        if (gp.pcount * (int64_t)(gp.aggs.size() + 1) > 8192) { F.why_not = "group domain too large for the LDS-resident grouped scan"; return F; }
        F.gscans.push_back(gp);
    }
    for (int id : P.outputs) {
        const Sym &s = B.sym[(size_t)id];
        FusedOutput fo;
        fo.node = id; fo.value = s.sc;
        if (s.kind == Sym::FOLD) fo.scan = s.scan; else fo.gscan = s.scan;
        F.outputs.push_back(fo);
    }
    F.ok = true;
    return F;
}

static void show_scalar(const Scalar &s, std::ostringstream &o) {
    switch (s.k) {
    case Scalar::AGG: if (s.agg < 0) o << "count"; else o << "agg" << s.agg; break;
    case Scalar::CONST: o << s.c; break;
    default: o << kBinNames[s.bin] << "("; show_scalar(*s.l, o); o << ","; show_scalar(*s.r, o); o << ")"; break;
    }
}

// Symbolic run over the key steps: each accumulator is a list of components; a right shift / subtract applies to a
// lone unshifted component, a left shift to all of them (it distributes over the OR), a final mask is kept aside.
int composite_key(const KeyStep *steps, int n, KeyComp *comps, int *masked_out, int64_t *mask_out) {
    if (masked_out) *masked_out = 0;
    if (mask_out) *mask_out = 0;
    if (getenv("VDL_NO_CANON_KEY") || n <= 0) return 0;
    struct Form { KeyComp c[kMaxKeyComps]; int n = 0; bool shifted[kMaxKeyComps] = {}, subbed[kMaxKeyComps] = {}; };
    Form f[2];
    bool masked = false;
    int64_t mask = 0;
    for (int s = 0; s < n; s++) {
        const KeyStep &st = steps[s];
        if (masked) return 0;                                          // the mask must be the last step
        if (st.kind == KeyStep::LOAD) {
            if (st.target < 0 || st.target > 1 || st.col < 0) return 0;
            Form &t = f[st.target];
            t = Form{};
            t.n = 1; t.c[0].col = st.col;
        } else if (st.kind == KeyStep::OPK) {
            if (st.target < 0 || st.target > 1) return 0;
            Form &t = f[st.target];
            if (t.n < 1 || st.const_left) return 0;
            if (st.bin == B_SHIFT && st.k >= 0) {                      // right shift: first thing done to a loaded column
                if (t.n != 1 || t.shifted[0] || t.subbed[0] || t.c[0].lsh != 0 || st.k > 63) return 0;
                t.c[0].rsh = (int)st.k; t.shifted[0] = true;
            } else if (st.bin == B_SHIFT) {                            // left shift: distributes over the OR of components
                if (st.k < -63) return 0;
                for (int k = 0; k < t.n; k++) { t.c[k].lsh += (int)(-st.k); if (t.c[k].lsh > 63) return 0; }
            } else if (st.bin == B_SUB || st.bin == B_ADD) {
                if (t.n != 1 || t.c[0].lsh != 0) return 0;
                t.c[0].sub = (int64_t)((uint64_t)t.c[0].sub + (st.bin == B_SUB ? (uint64_t)st.k : (uint64_t)0 - (uint64_t)st.k));
                t.subbed[0] = true;
            } else if (st.bin == B_BAND && st.target == 0) {
                masked = true; mask = st.k;
            } else {
                return 0;
            }
        } else {                                                       // COMBINE: acc = acc | tmp
            if (st.bin != B_BOR || f[0].n < 1 || f[1].n < 1 || f[0].n + f[1].n > kMaxKeyComps) return 0;
            for (int k = 0; k < f[1].n; k++) { f[0].c[f[0].n] = f[1].c[k]; f[0].n++; }
            f[1] = Form{};
        }
    }
    if (f[0].n < 1) return 0;
    for (int k = 0; k < f[0].n; k++) comps[k] = f[0].c[k];
    if (masked_out) *masked_out = masked ? 1 : 0;
    if (mask_out) *mask_out = mask;
    return f[0].n;
}

std::string describe_fused(const FusedPlan &F) {
    std::ostringstream o;
    if (!F.ok) { o << "not fused: " << F.why_not << "\n"; return o.str(); }
    for (size_t i = 0; i < F.scans.size(); i++) {
        const ScanPlan &sp = F.scans[i];
        o << "scan " << i << " table=" << sp.table << (sp.never ? " [never]" : "") << "\n";
        for (size_t c = 0; c < sp.cols.size(); c++) {
            o << "  col " << c << " " << sp.cols[c].name;
            if (sp.cols[c].lo != INT64_MIN || sp.cols[c].hi != INT64_MAX) {
                o << " in [";
                if (sp.cols[c].lo == INT64_MIN) o << "-inf"; else o << sp.cols[c].lo;
                o << ",";
                if (sp.cols[c].hi == INT64_MAX) o << "+inf"; else o << sp.cols[c].hi;
                o << "]";
            }
            o << "\n";
        }
        for (size_t a = 0; a < sp.aggs.size(); a++) {
            const ScanAgg &ag = sp.aggs[a];
            o << "  agg" << a << " " << (ag.kind == AGG_SUM ? "sum" : ag.kind == AGG_MIN ? "min" : "max") << " ";
            if (ag.fac.empty()) o << ag.constant;
            for (size_t f = 0; f < ag.fac.size(); f++) {
                if (f) o << " * ";
                o << "(" << ag.fac[f].a << " + " << ag.fac[f].s << "*col" << ag.fac[f].col << ")";
            }
            o << "\n";
        }
    }
    for (size_t i = 0; i < F.gscans.size(); i++) {
        const GroupScanPlan &gp = F.gscans[i];
        o << "group-scan " << i << " table=" << gp.table << " buckets=[" << gp.pmin << "," << gp.pmin + gp.pcount - 1 << "]"
          << (gp.never ? " [never]" : "") << "\n";
        for (size_t c = 0; c < gp.cols.size(); c++) {
            o << "  col " << c << " " << gp.cols[c].name;
            if (gp.cols[c].lo != INT64_MIN || gp.cols[c].hi != INT64_MAX) {
                o << " in [";
                if (gp.cols[c].lo == INT64_MIN) o << "-inf"; else o << gp.cols[c].lo;
                o << ",";
                if (gp.cols[c].hi == INT64_MAX) o << "+inf"; else o << gp.cols[c].hi;
                o << "]";
            }
            o << "\n";
        }
        o << "  key:";
        for (const KeyStep &k : gp.key) {
            const char *t = k.target ? "tmp" : "acc";
            if (k.kind == KeyStep::LOAD) o << " " << t << "=col" << k.col << ";";
            else if (k.kind == KeyStep::OPK) { if (k.const_left) o << " " << t << "=" << kBinNames[k.bin] << "(" << k.k << "," << t << ");"; else o << " " << t << "=" << kBinNames[k.bin] << "(" << t << "," << k.k << ");"; }
            else o << (k.const_left ? " acc=" : " acc=") << kBinNames[k.bin] << (k.const_left ? "(tmp,acc);" : "(acc,tmp);");
        }
        o << "\n";
        {
            KeyComp kc[kMaxKeyComps];
            int masked = 0;
            int64_t mask = 0;
            const int nc = composite_key(gp.key.data(), (int)gp.key.size(), kc, &masked, &mask);
            if (nc > 0) {
                o << "  key form: composite,";
                for (int k = 0; k < nc; k++) o << (k ? " |" : "") << " ((col" << kc[k].col << " >> " << kc[k].rsh << ") - " << kc[k].sub << ") << " << kc[k].lsh;
                if (masked) o << ", & " << mask;
                o << " (straight-line code)\n";
            } else {
                o << "  key form: general (interpreted step by step)\n";
            }
        }
        for (size_t a = 0; a < gp.aggs.size(); a++) {
            const ScanAgg &ag = gp.aggs[a];
            o << "  agg" << a << " " << (ag.kind == AGG_SUM ? "sum" : ag.kind == AGG_MIN ? "min" : ag.kind == AGG_MAX ? "max" : "first") << " ";
            if (ag.fac.empty()) o << ag.constant;
            for (size_t f = 0; f < ag.fac.size(); f++) {
                if (f) o << " * ";
                o << "(" << ag.fac[f].a << " + " << ag.fac[f].s << "*col" << ag.fac[f].col << ")";
            }
            o << "\n";
        }
    }
    for (auto &out : F.outputs) {
        if (out.gscan >= 0) o << "output Id " << out.node << " = group-scan " << out.gscan << " ";
        else o << "output Id " << out.node << " = scan " << out.scan << " ";
        show_scalar(*out.value, o);
        o << "\n";
    }
    return o.str();
}

}  // namespace vdl

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
#include <functional>
#include <set>
#include <sstream>

namespace vdl {
namespace {

struct Row;
using RowP = std::shared_ptr<const Row>;
struct Row {
    //  VIA  : l = an index expression over THIS table (a join-index column, possibly itself looked up), r = an
    //         expression over the rows of table `dim`: the value of r at dim row l (Gather through a foreign key);
    //         `dsel` = selection under which the dim-side vector holds values, `col` = a column of `dim` (its length)
    //  LIKE : l = argument (heap offsets), col = the heap column, pat = the pattern
    enum K { COL, CONST, IOTA, BIN, VIA, LIKE } k = CONST;
    std::string col, dim, pat;
    int dsel = 0;
    int64_t c0 = 0, c1 = 0;  // CONST: value; IOTA: from, step
    int bin = -1;
    RowP l, r;
};

RowP mk_col(const std::string &c) { auto p = std::make_shared<Row>(); p->k = Row::COL; p->col = c; return p; }
RowP mk_const(int64_t v) { auto p = std::make_shared<Row>(); p->k = Row::CONST; p->c0 = v; return p; }
RowP mk_iota(int64_t f, int64_t s) { auto p = std::make_shared<Row>(); p->k = Row::IOTA; p->c0 = f; p->c1 = s; return p; }
static bool is_const(const RowP &e, int64_t v) { return e->k == Row::CONST && e->c0 == v; }
RowP mk_bin(int op, RowP l, RowP r) {
    if (l->k == Row::CONST && r->k == Row::CONST) return mk_const(apply_bin(op, l->c0, r->c0));
    // value identities (validity lives in the selection, not in the expression): the emitter multiplies by the
    // constants of CASE WHEN (Vlite.hs:240-245: cond * a + (1 - cond) * b with b = 0) and adds zeros
    if (op == B_MUL) {
        if (is_const(l, 0) || is_const(r, 0)) return mk_const(0);
        if (is_const(l, 1)) return r;
        if (is_const(r, 1)) return l;
    }
    if (op == B_ADD && is_const(l, 0)) return r;
    if ((op == B_ADD || op == B_SUB) && is_const(r, 0)) return l;
    auto p = std::make_shared<Row>(); p->k = Row::BIN; p->bin = op; p->l = std::move(l); p->r = std::move(r); return p;
}
RowP mk_via(RowP idx, RowP inner, const std::string &dim, int dsel, const std::string &len_col) {
    auto p = std::make_shared<Row>(); p->k = Row::VIA; p->l = std::move(idx); p->r = std::move(inner); p->dim = dim; p->dsel = dsel; p->col = len_col; return p;
}
RowP mk_like(RowP arg, const std::string &heap, const std::string &pat) {
    auto p = std::make_shared<Row>(); p->k = Row::LIKE; p->l = std::move(arg); p->col = heap; p->pat = pat; return p;
}

// canonical text of an expression: structural identity (selections, virtual columns)
void row_key(const RowP &e, std::string &o) {
    switch (e->k) {
    case Row::COL: o += e->col; break;
    case Row::CONST: o += "#" + std::to_string(e->c0); break;
    case Row::IOTA: o += "iota(" + std::to_string(e->c0) + "," + std::to_string(e->c1) + ")"; break;
    case Row::BIN: o += kBinNames[e->bin]; o += "("; row_key(e->l, o); o += ","; row_key(e->r, o); o += ")"; break;
    case Row::VIA: o += "via(" + e->dim + ":" + std::to_string(e->dsel) + ";"; row_key(e->l, o); o += ";"; row_key(e->r, o); o += ")"; break;
    case Row::LIKE: o += "like(" + e->col + ";" + e->pat + ";"; row_key(e->l, o); o += ")"; break;
    }
}
std::string row_key(const RowP &e) { std::string o; row_key(e, o); return o; }

// Atoms = what a scan can hold in a register per row: a table column, a value looked up through an index atom (VIA of
// a plain dim column), a LIKE over an atom of heap offsets, or the difference of two atoms (column against column).
bool is_atom(const RowP &e) {
    switch (e->k) {
    case Row::COL: return true;
    case Row::VIA: return (is_atom(e->l) || (e->l->k == Row::IOTA && e->l->c0 == 0 && e->l->c1 == 1 && is_const(e->r, 1))) && (e->r->k == Row::COL || is_const(e->r, 1));
    case Row::LIKE: return is_atom(e->l);
    case Row::BIN: return e->bin == B_SUB && is_atom(e->l) && is_atom(e->r);
    default: return false;
    }
}
bool is_boolean_atom(const RowP &e) { return e->k == Row::LIKE || (e->k == Row::VIA && is_const(e->r, 1)); }

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
    //  POSSET : Scatter(constant, _, index [mod N]) over the selected rows of `table` -- a set of positions P (a vector holding
    //           the constant at p in P, EPS elsewhere); r_step = 1: RangeV 0 1 / FoldSelect of it (p at p in P).  e = the index
    //           atom, r_count = N (0: no mod), sel = the source rows.  Only ever consumed as Gather positions: a semi-join.
    enum Kind { NONE, ROW, FOLD, RANGEC, PART, SORTED, GFOLD, POSSET } kind = NONE;
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
    case Row::BIN: return a->bin == b->bin && row_equal(a->l, b->l) && row_equal(a->r, b->r);
    default: return row_key(a) == row_key(b);
    }
}

struct Selection { std::string table; RowP pred; std::string key; };   // pred == nullptr: all rows; key: canonical conjunct set

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
// (`cols` is keyed by the canonical text of an atom; `atoms` keeps the atoms themselves for the lowering)
struct Clause { std::map<std::string, IvSet> cols; std::map<std::string, RowP> atoms; bool never = false; };

static void clause_and(Clause &out, const RowP &atom, const IvSet &set) {
    const std::string key = row_key(atom);
    out.atoms[key] = atom;
    auto it = out.cols.find(key);
    if (it == out.cols.end()) out.cols[key] = iv_norm(set); else it->second = iv_and(it->second, set);
}

bool leaf_cmp(const Row &p, RowP &atom, IvSet &set) {
    if (p.k != Row::BIN) return false;
    const RowP &l = p.l, &r = p.r;
    if (p.bin == B_GT) {
        if (is_atom(l) && r->k == Row::CONST) {           // col > k
            atom = l; set = r->c0 == INT64_MAX ? IvSet{} : IvSet{{r->c0 + 1, INT64_MAX}}; return true;
        }
        if (l->k == Row::CONST && is_atom(r)) {           // k > col
            atom = r; set = l->c0 == INT64_MIN ? IvSet{} : IvSet{{INT64_MIN, l->c0 - 1}}; return true;
        }
        if (is_atom(l) && is_atom(r)) {                   // column against column: a > b  <=>  a - b >= 1 (no wrap-around for the
            atom = mk_bin(B_SUB, l, r); set = IvSet{{1, INT64_MAX}}; return true;      // date / key / quantity magnitudes compared this way)
        }
    } else if (p.bin == B_EQ) {
        if (is_atom(l) && r->k == Row::CONST) { atom = l; set = {{r->c0, r->c0}}; return true; }
        if (l->k == Row::CONST && is_atom(r)) { atom = r; set = {{l->c0, l->c0}}; return true; }
        if (is_atom(l) && is_atom(r)) { atom = mk_bin(B_SUB, l, r); set = {{0, 0}}; return true; }
    }
    return false;
}

// A 0 / 1 valued expression built from range tests of atoms with AND / OR / NOT (NOT as the emitter writes it: Equals(x, 0),
// Subtract(1, x): Vlite.hs:240-245).  What does not reduce to per-column ranges is evaluated per row as a formula column
// (VC_FORM): disjunctions across columns, CASE WHEN conditions that are aggregate inputs.
bool is_formula(const RowP &e) {
    if (e->k == Row::CONST) return e->c0 == 0 || e->c0 == 1;
    if (is_boolean_atom(e)) return true;
    if (e->k != Row::BIN) return false;
    RowP a; IvSet s;
    if (leaf_cmp(*e, a, s)) return true;
    if (e->bin == B_LAND || e->bin == B_LOR) return is_formula(e->l) && is_formula(e->r);
    if (e->bin == B_EQ) return (is_const(e->r, 0) && is_formula(e->l)) || (is_const(e->l, 0) && is_formula(e->r));
    if (e->bin == B_SUB) return is_const(e->l, 1) && is_formula(e->r);
    return false;
}

bool to_clause_strict(const RowP &p, Clause &out);
// predicate "p != 0" -> Clause: per-atom interval sets, conjuncts of any other boolean shape as formula "atoms" that must be 1;
// false when p has a shape the scan kernels cannot express
bool to_clause(const RowP &p, Clause &out) {
    if (!p) return true;
    if (p->k == Row::BIN && p->bin == B_LAND) return to_clause(p->l, out) && to_clause(p->r, out);
    Clause tmp;
    if (to_clause_strict(p, tmp)) {
        out.never |= tmp.never;
        for (auto &kv : tmp.cols) clause_and(out, tmp.atoms.at(kv.first), kv.second);
        return true;
    }
    if (is_formula(p)) { clause_and(out, p, IvSet{{1, 1}}); return true; }
    return false;
}
bool to_clause_strict(const RowP &p, Clause &out) {
    if (!p) return true;
    if (p->k == Row::CONST) { if (p->c0 == 0) out.never = true; return true; }
    if (is_atom(p)) {                                      // a value read as a truth value (also: a lookup's validity, a LIKE)
        if (is_boolean_atom(p)) clause_and(out, p, IvSet{{1, 1}});
        else clause_and(out, p, IvSet{{INT64_MIN, -1}, {1, INT64_MAX}});
        return true;
    }
    if (p->k != Row::BIN) return false;
    RowP atom; IvSet set;
    if (leaf_cmp(*p, atom, set)) { clause_and(out, atom, set); return true; }
    // `k - row id` read as a truth value: every row but row k.  (What the reference's anti-join comes to: `ones - selectmask` with
    // selectmask re-bound to a vector of positions, /root/reference/src/Vlite.hs:1199-1229 -- SURVEY.md section 10; TPC-H Q16.)
    if (p->bin == B_SUB && p->l->k == Row::CONST && p->r->k == Row::IOTA && p->r->c0 == 0 && p->r->c1 == 1) {
        const int64_t k = p->l->c0;
        IvSet others;
        if (k > INT64_MIN) others.push_back({INT64_MIN, k - 1});
        if (k < INT64_MAX) others.push_back({k + 1, INT64_MAX});
        clause_and(out, p->r, others);
        return true;
    }
    if (p->bin == B_LAND) return to_clause_strict(p->l, out) && to_clause_strict(p->r, out);
    if (p->bin == B_LOR) {
        // only a disjunction of conditions on ONE column stays a per-column filter
        // (e.g. `<=` printed as LogicalOr(Greater, Equals), /root/reference/src/Vdl.hs:143-144)
        Clause a, b;
        if (!to_clause_strict(p->l, a) || !to_clause_strict(p->r, b)) return false;
        if (a.never && b.never) { out.never = true; return true; }
        if (a.never) { for (auto &kv : b.cols) clause_and(out, b.atoms.at(kv.first), kv.second); return true; }
        if (b.never) { for (auto &kv : a.cols) clause_and(out, a.atoms.at(kv.first), kv.second); return true; }
        if (a.cols.empty() || b.cols.empty()) return true;   // one side is constant true
        if (a.cols.size() != 1 || b.cols.size() != 1 || a.cols.begin()->first != b.cols.begin()->first) return false;
        clause_and(out, a.atoms.begin()->second, iv_or(a.cols.begin()->second, b.cols.begin()->second));
        return true;
    }
    return false;
}

// ---- aggregate data: product of affine single-column factors ----------------------
struct Affine { bool has_col = false; std::string col; int64_t a = 0, s = 0; RowP atom; };   // a + s*atom (col = the atom's canonical text)

bool to_affine(const RowP &e, Affine &out) {
    if (e->k == Row::CONST) { out = Affine{false, "", e->c0, 0, nullptr}; return true; }
    if (e->k != Row::BIN || e->bin != B_SUB) {             // (a difference of atoms is affine algebra, not a SUB atom, when it can be)
        if (is_atom(e)) { out = Affine{true, row_key(e), 0, 1, e}; return true; }
    }
    if (e->k != Row::BIN) return false;
    Affine x, y;
    // NOT of a 0 / 1 atom, as the emitter writes it: Equals(x, 0) (negcond = cond ==. zeros, Vlite.hs:240)
    auto boolean = [](const RowP &x) { return is_boolean_atom(x) || (x->k == Row::BIN && is_formula(x)); };
    if (e->bin == B_EQ && ((is_const(e->r, 0) && boolean(e->l)) || (is_const(e->l, 0) && boolean(e->r)))) {
        const RowP &b = is_const(e->r, 0) ? e->l : e->r;
        out = Affine{true, row_key(b), 1, -1, b};
        return true;
    }
    // a condition as a number (CASE WHEN c THEN x ELSE 0 is c * x): a formula column
    if ((e->bin == B_LAND || e->bin == B_LOR || e->bin == B_GT || e->bin == B_EQ) && is_formula(e)) {
        out = Affine{true, row_key(e), 0, 1, e};
        return true;
    }
    if (e->bin == B_ADD || e->bin == B_SUB) {
        if (!to_affine(e->l, x) || !to_affine(e->r, y)) return false;
        if (x.has_col && y.has_col && x.col != y.col) return false;
        int64_t sign = e->bin == B_ADD ? 1 : -1;
        out.has_col = x.has_col || y.has_col;
        out.col = x.has_col ? x.col : y.col;
        out.atom = x.has_col ? x.atom : y.atom;
        out.a = apply_bin(B_ADD, x.a, apply_bin(B_MUL, sign, y.a));
        out.s = apply_bin(B_ADD, x.s, apply_bin(B_MUL, sign, y.s));
        return true;
    }
    if (e->bin == B_MUL) {
        if (!to_affine(e->l, x) || !to_affine(e->r, y)) return false;
        if (x.has_col && y.has_col) return false;          // quadratic: handled as two factors
        const Affine &k = x.has_col ? y : x, &v = x.has_col ? x : y;
        out.has_col = v.has_col; out.col = v.col; out.atom = v.atom;
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

    std::map<std::string, int> sel_intern;                     // canonical conjunct set -> selection index
    std::map<int, int> witness;                                // selection -> first statement whose vector is EPS exactly outside it
    struct SemiSet { std::string table; int sel; RowP index; int64_t modulus; };
    std::vector<SemiSet> semis;                                // position sets consumed as semi-joins: VIA::dsel = -(k + 1)
    int semi_for(const Sym &ps) {
        for (size_t k = 0; k < semis.size(); k++)
            if (semis[k].table == ps.table && semis[k].sel == ps.sel && semis[k].modulus == ps.r_count && row_equal(semis[k].index, ps.e)) return (int)k;
        semis.push_back({ps.table, ps.sel, ps.e, ps.r_count});
        return (int)semis.size() - 1;
    }
    std::map<std::string, std::string> table_col;              // table -> one of its columns (for the table's length)

    // A selection is a SET of conjuncts: two routes to the same filter (the emitter gathers every column of a join
    // separately, each gather adding the same validity conditions) must be one selection, or their folds would land
    // in different scans.  Conjunctions are flattened, constants dropped, duplicates merged, the rest ordered by text.
    static void flatten(const RowP &p, std::map<std::string, RowP> &out, bool &never) {
        if (!p) return;
        if (p->k == Row::CONST) { if (p->c0 == 0) never = true; return; }
        if (p->k == Row::BIN && p->bin == B_LAND) { flatten(p->l, out, never); flatten(p->r, out, never); return; }
        out[row_key(p)] = p;
    }
    int new_sel(const std::string &table, RowP pred) {
        std::map<std::string, RowP> conj;
        bool never = false;
        flatten(pred, conj, never);
        if (never) { conj.clear(); conj["#0"] = mk_const(0); }
        if (conj.empty()) return 0;                             // constant true: every row
        std::string key = table + "|";
        RowP all;
        for (auto &kv : conj) {
            key += kv.first + "&";
            if (!all) all = kv.second;
            else { auto n = std::make_shared<Row>(); n->k = Row::BIN; n->bin = B_LAND; n->l = all; n->r = kv.second; all = n; }
        }
        auto it = sel_intern.find(key);
        if (it != sel_intern.end()) return it->second;
        sels.push_back({table, all, key});
        sel_intern[key] = (int)sels.size();
        return (int)sels.size();
    }
    RowP pred_of(int s) const { return s ? sels[(size_t)s - 1].pred : nullptr; }

    int combine_sel(const std::string &table, int a, int b) {
        if (a == b || b == 0) return a;
        if (a == 0) return b;
        auto key = std::make_pair(std::min(a, b), std::max(a, b));
        auto it = conj_memo.find(key);
        if (it != conj_memo.end()) return it->second;
        auto both = std::make_shared<Row>(); both->k = Row::BIN; both->bin = B_LAND; both->l = pred_of(a); both->r = pred_of(b);
        int s = new_sel(table, both);
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
            const bool heap = n.column.size() > 5 && n.column.compare(n.column.size() - 5, 5, ".heap") == 0;
            if (heap) out.table = n.column;                     // a string heap is a vector of its own length, not a column of the table
            else if (!table_col.count(out.table)) table_col[out.table] = n.column;
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
            if (src.kind == Sym::ROW && pos.kind == Sym::ROW && fold.kind == Sym::ROW && src.table == pos.table && fold.table == pos.table &&
                src.e->k == Row::CONST && src.e->c0 != 0) {
                // a constant scattered by an index (mod N): duplicates write the same value -- the set of positions (semi-join,
                // Vlite.hs:1212-1222)
                RowP index = pos.e;
                int64_t modulus = 0;
                if (index->k == Row::BIN && index->bin == B_MOD && index->r->k == Row::CONST && index->r->c0 > 0) { modulus = index->r->c0; index = index->l; }
                if (is_atom(index)) {
                    out.kind = Sym::POSSET; out.table = src.table; out.e = index;
                    out.sel = combine_sel(src.table, src.sel, pos.sel);
                    out.r_from = src.e->c0; out.r_count = modulus; out.r_step = 0;
                    return out;
                }
            }
            if (src.kind == Sym::ROW && pos.kind == Sym::ROW && fold.kind == Sym::ROW && src.table == pos.table && fold.table == pos.table &&
                pos.e->k == Row::IOTA && pos.e->c0 == 0 && pos.e->c1 == 1) {
                // positions = the slots' own ids (the dimension side of a join scatters ones / row ids back by
                // Gather(rowids, FoldSelect(..)), Vlite.hs:1268-1275): the source, restricted to those slots
                out.kind = Sym::ROW; out.table = src.table; out.e = src.e;
                out.sel = combine_sel(src.table, src.sel, pos.sel);
                return out;
            }
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
            } else if (r.kind == Sym::POSSET && n.imm0 == 0 && n.imm1 == 1) {
                out = r; out.r_step = 1;                                // the positions themselves
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
            if (ctl.kind == Sym::POSSET && d.kind == Sym::POSSET && ctl.r_step == 1 && d.r_step == 0 && d.r_from != 0 && ctl.table == d.table &&
                ctl.sel == d.sel && ctl.r_count == d.r_count && row_equal(ctl.e, d.e)) {
                out = ctl;                                              // FoldSelect(RangeV 0 1 set, set): p where the set holds its (non-zero) constant
                return out;
            }
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
                        if (cl.atoms.at(kv.first)->k != Row::COL) { fits = false; break; }     // k_filter_columns reads table columns only
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
            if (src.kind == Sym::ROW && pos.kind == Sym::POSSET && pos.r_step == 1 && table_col.count(src.table)) {
                // src[p] for p in the set: the rows of src's table that some selected row of the other table points at -- a
                // semi-join.  (The result has the other table's length; its slots beyond src's rows are EPS and nothing but
                // folds and compactions ever looks at it.)  As a row expression of src's table: src, restricted to the rows
                // whose own id is in the set.
                const int k = semi_for(pos);
                out.kind = Sym::ROW; out.table = src.table; out.e = src.e;
                out.sel = combine_sel(src.table, src.sel, new_sel(src.table, mk_via(mk_iota(0, 1), mk_const(1), "#semi" + std::to_string(k), -(k + 1), table_col.at(src.table))));
                return out;
            }
            if (src.kind != Sym::ROW || pos.kind != Sym::ROW) return out;
            if (src.table != pos.table) {
                // a lookup into another table's vector through an index expression: the FK-join lowering seen from the fact
                // side (Vlite.hs:1199-1282).  The result lives on pos' table; it is EPS where the index is EPS or out of
                // range or the looked-up slot is EPS: that condition joins the selection.
                auto len = table_col.find(src.table);
                if (len == table_col.end() || !is_atom(pos.e)) return out;
                out.kind = Sym::ROW; out.table = pos.table;
                out.sel = combine_sel(pos.table, pos.sel, new_sel(pos.table, mk_via(pos.e, mk_const(1), src.table, src.sel, len->second)));
                if (src.e->k == Row::IOTA && src.e->c0 == 0 && src.e->c1 == 1) out.e = pos.e;       // row ids looked up by row id
                else if (src.e->k == Row::CONST) out.e = src.e;
                else out.e = mk_via(pos.e, src.e, src.table, src.sel, len->second);
                return out;
            }
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
        case Op::Like: {
            // Like over a vector of heap offsets and the (unfiltered) heap itself, Vdl.hs:444-447
            const Sym &d = S(n.a), &heap = S(n.b);
            if (d.kind != Sym::ROW || heap.kind != Sym::ROW || heap.e->k != Row::COL || heap.sel != 0) return out;
            out.kind = Sym::ROW; out.table = d.table; out.sel = d.sel;
            out.e = mk_like(d.e, heap.e->col, n.pattern);
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
    default: o << row_key(e); break;
    }
}


// key expression -> two-accumulator program (vdl_fuse.h KeyStep); false if the tree is not
// left/right-deep with single-column leaves
bool single_column_chain(const RowP &e) {
    if (is_atom(e) && e->k != Row::BIN) return true;
    if (e->k != Row::BIN) return false;
    if (e->r->k == Row::CONST) return single_column_chain(e->l);
    if (e->l->k == Row::CONST) return single_column_chain(e->r);
    return false;
}

template <typename ColIndex>
bool emit_key(const RowP &e, int target, std::vector<KeyStep> &prog, ColIndex &col_index) {
    KeyStep st;
    if (is_atom(e) && e->k != Row::BIN) { st.kind = KeyStep::LOAD; st.target = target; st.col = col_index(e); if (st.col < 0) return false; prog.push_back(st); return true; }
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

// Atoms -> the scan's virtual columns (vdl_fuse.h VColKind); dependencies are appended before their dependents, so a
// column only ever refers to earlier ones.  -1 with `why` set when an atom has no column form.
struct VCols {
    std::vector<ScanColumn> &cols;
    FusedPlan &F;
    const Builder &B;
    std::string &why;
    std::map<std::string, int> index;                   // canonical text -> column

    int prelude_bitmap(int dsel) {
        auto w = B.witness.find(dsel);
        if (w == B.witness.end()) { why = "no statement holds the dimension-side selection " + std::to_string(dsel); return -1; }
        for (size_t k = 0; k < F.prelude.size(); k++)
            if (F.prelude[k].kind == PreludeItem::DIM_BITMAP && F.prelude[k].witness == w->second) return (int)k;
        PreludeItem it; it.kind = PreludeItem::DIM_BITMAP; it.witness = w->second;
        lower_dimension(dsel, it);                              // (may append the items ITS lookups need: they come first)
        F.prelude.push_back(it);
        return (int)F.prelude.size() - 1;
    }
    // formula -> postfix steps; the columns of its leaves are appended first
    void push_leaf(std::vector<FormStep> &out, int col, const IvSet &set, int &depth, int &deepest) {
        if (set.empty()) out.push_back(FormStep{FormStep::FALSE_, -1, 0, 0});
        for (size_t i = 0; i < set.size(); i++) {
            out.push_back(FormStep{FormStep::LEAF, col, set[i].first, set[i].second});
            if (i) out.push_back(FormStep{FormStep::OR, -1, 0, 0});
        }
        depth += 1; deepest = std::max(deepest, depth + (set.size() > 1 ? 1 : 0));
    }
    bool compile_formula(const RowP &e, std::vector<FormStep> &out, int &depth, int &deepest) {
        if (deepest >= kMaxFormDepth) { why = "condition nested too deeply"; return false; }
        if (e->k == Row::CONST) { out.push_back(FormStep{e->c0 ? FormStep::TRUE_ : FormStep::FALSE_, -1, 0, 0}); depth++; deepest = std::max(deepest, depth); return true; }
        {   // a sub-formula over ONE atom: a single leaf with that atom's interval set
            Clause one;
            if (to_clause_strict(e, one) && !one.never && one.cols.size() == 1) {
                const int col = (*this)(one.atoms.begin()->second);
                if (col < 0) return false;
                push_leaf(out, col, one.cols.begin()->second, depth, deepest);
                return true;
            }
        }
        if (is_boolean_atom(e)) {
            const int col = (*this)(e);
            if (col < 0) return false;
            push_leaf(out, col, IvSet{{1, 1}}, depth, deepest);
            return true;
        }
        if (e->k != Row::BIN) { why = "not a condition: " + row_key(e); return false; }
        RowP a; IvSet set;
        if (leaf_cmp(*e, a, set)) {
            const int col = (*this)(a);
            if (col < 0) return false;
            push_leaf(out, col, iv_norm(set), depth, deepest);
            return true;
        }
        if (e->bin == B_LAND) {
            // the conjuncts that are range tests merge per atom (q >= 100 and q <= 1100 is one leaf); the others are formulas
            std::vector<RowP> conj, rest;
            std::function<void(const RowP &)> flat = [&](const RowP &x) {
                if (x->k == Row::BIN && x->bin == B_LAND) { flat(x->l); flat(x->r); } else conj.push_back(x);
            };
            flat(e);
            Clause acc;
            for (const RowP &x : conj) {
                Clause t;
                if (to_clause_strict(x, t)) { acc.never |= t.never; for (auto &kv : t.cols) clause_and(acc, t.atoms.at(kv.first), kv.second); }
                else rest.push_back(x);
            }
            for (auto &kv : acc.cols) acc.never |= kv.second.empty();
            if (acc.never) { out.push_back(FormStep{FormStep::FALSE_, -1, 0, 0}); depth++; deepest = std::max(deepest, depth); return true; }
            int terms = 0;
            for (auto &kv : acc.cols) {
                const int col = (*this)(acc.atoms.at(kv.first));
                if (col < 0) return false;
                push_leaf(out, col, kv.second, depth, deepest);
                terms++;
            }
            for (const RowP &x : rest) { if (!compile_formula(x, out, depth, deepest)) return false; terms++; }
            if (terms == 0) { out.push_back(FormStep{FormStep::TRUE_, -1, 0, 0}); depth++; deepest = std::max(deepest, depth); return true; }
            for (int k = 1; k < terms; k++) { out.push_back(FormStep{FormStep::AND, -1, 0, 0}); depth--; }
            return true;
        }
        if (e->bin == B_LOR) {
            if (!compile_formula(e->l, out, depth, deepest) || !compile_formula(e->r, out, depth, deepest)) return false;
            out.push_back(FormStep{FormStep::OR, -1, 0, 0});
            depth--;
            return true;
        }
        const RowP *inner = nullptr;
        if (e->bin == B_EQ) inner = is_const(e->r, 0) ? &e->l : is_const(e->l, 0) ? &e->r : nullptr;
        else if (e->bin == B_SUB && is_const(e->l, 1)) inner = &e->r;
        if (!inner) { why = "not a condition: " + row_key(e); return false; }
        if (!compile_formula(*inner, out, depth, deepest)) return false;
        out.push_back(FormStep{FormStep::NOT, -1, 0, 0});
        return true;
    }
    // filters of a clause -> range filters of the columns; an interval SET on one atom (IN lists) -> a formula column that must be 1
    bool lower_filters(const Clause &cl, bool &never) {
        for (auto &kv : cl.cols) {
            const RowP &atom = cl.atoms.at(kv.first);
            const int c = (*this)(atom);
            if (c < 0) return false;
            if (kv.second.empty()) { never = true; continue; }
            if (kv.second.size() == 1) {
                cols[(size_t)c].lo = std::max(cols[(size_t)c].lo, kv.second[0].first);
                cols[(size_t)c].hi = std::min(cols[(size_t)c].hi, kv.second[0].second);
                continue;
            }
            std::string key = "in(" + kv.first;
            for (auto &iv : kv.second) key += ";" + std::to_string(iv.first) + ".." + std::to_string(iv.second);
            key += ")";
            if (!index.count(key)) {
                ScanColumn f;
                f.kind = VC_FORM;
                int depth = 0, deepest = 0;
                push_leaf(f.form, c, kv.second, depth, deepest);
                if ((int)f.form.size() > kMaxFormSteps) { why = "filter on " + kv.first + " has too many ranges"; return false; }
                cols.push_back(f);
                index[key] = (int)cols.size() - 1;
            }
            ScanColumn &f = cols[(size_t)index[key]];
            f.lo = 1; f.hi = 1;
        }
        return true;
    }
    // the dimension-side selection as a scan of its own (PreludeItem::scan), when it has the form
    void lower_dimension(int dsel, PreludeItem &it) {
        const Selection &S = B.sels[(size_t)dsel - 1];
        Clause cl;
        if (!to_clause(S.pred, cl)) return;
        std::string inner_why;
        std::vector<ScanColumn> dc;
        VCols inner{dc, F, B, inner_why, {}};
        bool never = cl.never;
        if (!inner.lower_filters(cl, never)) return;
        tidy_columns(dc, {});
        if ((int)dc.size() > kMaxSelectCols) return;
        bool direct = false;
        for (const ScanColumn &c : dc) direct |= c.kind == VC_DIRECT && c.name.compare(0, S.table.size() + 1, S.table + ".") == 0;
        if (!direct) return;
        it.scan = true; it.table = S.table; it.cols = dc; it.never = never;
    }
    // Last step of a lowering: drops range checks that a lookup through the same index column performs anyway, and orders the
    // columns so that the scan kernels (which derive columns in index order and fold a filter the moment its column exists,
    // so that later lookups only go out for rows still alive) meet the cheap deciding columns first: table columns, then
    // differences / conditions over them, bitmap lookups, and last the lookups of dimension columns.  Returns old -> new
    // index (-1: dropped) for whoever refers to columns by index.
    static std::vector<int> tidy_columns(std::vector<ScanColumn> &cols, const std::vector<int> &keep) {
        const size_t n = cols.size();
        std::vector<char> drop(n, 0);
        for (size_t k = 0; k < n; k++) {
            if (cols[k].kind != VC_INRANGE) continue;
            bool redundant = false;
            for (const ScanColumn &o : cols) redundant |= (o.kind == VC_BITS || o.kind == VC_GATHER) && o.idx == cols[k].idx;
            for (int kc : keep) if (kc == (int)k) redundant = false;
            for (const ScanColumn &o : cols) for (int src : o.sources()) if (src == (int)k) redundant = false;
            drop[k] = redundant;
        }
        // does the column, or something computed from it, carry a filter?
        std::vector<char> drives(n, 0);
        for (size_t k = n; k-- > 0;) {
            if (drop[k]) continue;
            if (cols[k].lo != INT64_MIN || cols[k].hi != INT64_MAX || cols[k].kind == VC_INRANGE) drives[k] = 1;
            if (drives[k]) for (int src : cols[k].sources()) drives[(size_t)src] = 1;
        }
        auto cost = [&](size_t k) {
            switch (cols[k].kind) {
            case VC_DIRECT: return 0;
            case VC_SUB: case VC_FORM: case VC_ROWID: return 1;
            case VC_BITS: case VC_INRANGE: return 2;
            default: return 3;
            }
        };
        std::vector<int> map(n, -1), order;
        std::vector<char> placed(n, 0);
        for (size_t round = 0; round < n; round++) {
            int best = -1;
            for (size_t k = 0; k < n; k++) {
                if (placed[k] || drop[k]) continue;
                bool ready = true;
                for (int src : cols[k].sources()) ready &= (bool)placed[(size_t)src];
                if (!ready) continue;
                auto key = [&](size_t c) { return std::make_tuple(cols[c].kind == VC_DIRECT ? 0 : 1, drives[c] ? 0 : 1, cost(c), (int)c); };
                if (best < 0 || key(k) < key((size_t)best)) best = (int)k;
            }
            if (best < 0) break;
            placed[(size_t)best] = 1;
            map[(size_t)best] = (int)order.size();
            order.push_back(best);
        }
        std::vector<ScanColumn> out;
        for (int k : order) {
            ScanColumn c = cols[(size_t)k];
            if (c.idx >= 0) c.idx = map[(size_t)c.idx];
            if (c.idx2 >= 0) c.idx2 = map[(size_t)c.idx2];
            for (FormStep &f : c.form) if (f.op == FormStep::LEAF) f.col = map[(size_t)f.col];
            out.push_back(c);
        }
        cols = out;
        return map;
    }
    // the position set B.semis[k] as a scan of its table (PreludeItem::SEMI_BITMAP)
    int prelude_semi(int k, const std::string &bits_of) {
        const Builder::SemiSet &S = B.semis[(size_t)k];
        PreludeItem it; it.kind = PreludeItem::SEMI_BITMAP; it.scan = true; it.table = S.table; it.modulus = S.modulus; it.bits_of = bits_of;
        it.witness = -(k + 1);
        for (size_t j = 0; j < F.prelude.size(); j++)
            if (F.prelude[j].kind == PreludeItem::SEMI_BITMAP && F.prelude[j].witness == it.witness && F.prelude[j].bits_of == bits_of) return (int)j;
        Clause cl;
        if (!to_clause(B.pred_of(S.sel), cl)) { why = "the semi-join's source selection is not a conjunction of ranges / conditions"; return -1; }
        std::string inner_why;
        VCols inner{it.cols, F, B, inner_why, {}};
        it.never = cl.never;
        if (!inner.lower_filters(cl, it.never)) { why = "semi-join source: " + inner_why; return -1; }
        int index = inner(S.index);
        if (index < 0) { why = "semi-join index: " + inner_why; return -1; }
        const std::vector<int> map = tidy_columns(it.cols, {index});
        it.index_col = map[(size_t)index];
        if ((int)it.cols.size() > kMaxSelectCols) { why = "semi-join source scan needs more than " + std::to_string(kMaxSelectCols) + " columns"; return -1; }
        bool direct = false;
        for (const ScanColumn &c : it.cols) direct |= c.kind == VC_DIRECT && c.name.compare(0, S.table.size() + 1, S.table + ".") == 0;
        if (!direct) { why = "semi-join source scan touches no column of its table"; return -1; }
        F.prelude.push_back(it);
        return (int)F.prelude.size() - 1;
    }
    int prelude_lut(const std::string &heap, const std::string &pattern) {
        for (size_t k = 0; k < F.prelude.size(); k++)
            if (F.prelude[k].kind == PreludeItem::LIKE_LUT && F.prelude[k].heap == heap && F.prelude[k].pattern == pattern) return (int)k;
        PreludeItem it; it.kind = PreludeItem::LIKE_LUT; it.heap = heap; it.pattern = pattern;
        F.prelude.push_back(it);
        return (int)F.prelude.size() - 1;
    }
    int operator()(const RowP &atom) {
        const std::string key = row_key(atom);
        auto it = index.find(key);
        if (it != index.end()) return it->second;
        ScanColumn c;
        switch (atom->k) {
        case Row::COL: c.kind = VC_DIRECT; c.name = atom->col; break;
        case Row::IOTA: c.kind = VC_ROWID; break;          // the row's own id (index of a semi-join bit: is_atom)
        case Row::VIA: {
            c.idx = (*this)(atom->l);
            if (c.idx < 0) return -1;
            if (atom->dsel < 0) {                           // bit `own row id` of a position set built by a scan of another table
                c.kind = VC_BITS; c.prelude = prelude_semi(-atom->dsel - 1, atom->col);
                if (c.prelude < 0) return -1;
                break;
            }
            if (atom->r->k == Row::COL && atom->dsel == 0) { c.kind = VC_GATHER; c.name = atom->r->col; }
            else if (is_const(atom->r, 1) && atom->dsel == 0) { c.kind = VC_INRANGE; c.name = atom->col; }
            else if (is_const(atom->r, 1)) { c.kind = VC_BITS; c.prelude = prelude_bitmap(atom->dsel); if (c.prelude < 0) return -1; }
            else if (atom->r->k == Row::COL) {          // a column of a FILTERED dimension vector: the filter is a condition of its own
                c.kind = VC_GATHER; c.name = atom->r->col;    // (the selection carries via(..;#1) with the same dsel: see Gather in visit)
            } else { why = "value looked up through a foreign key is not a plain column: " + key; return -1; }
            break;
        }
        case Row::LIKE:
            c.idx = (*this)(atom->l);
            if (c.idx < 0) return -1;
            c.kind = VC_LUT; c.prelude = prelude_lut(atom->col, atom->pat);
            break;
        case Row::BIN:
            if (atom->bin == B_SUB && is_atom(atom->l) && is_atom(atom->r)) {      // SUB of two atoms
                c.idx = (*this)(atom->l); c.idx2 = (*this)(atom->r);
                if (c.idx < 0 || c.idx2 < 0) return -1;
                c.kind = VC_SUB;
                break;
            }
            if (!is_formula(atom)) { why = "not a column form: " + key; return -1; }
            c.kind = VC_FORM;
            { int depth = 0, deepest = 0; if (!compile_formula(atom, c.form, depth, deepest)) return -1; }
            if ((int)c.form.size() > kMaxFormSteps) { why = "condition with more than " + std::to_string(kMaxFormSteps) + " steps"; return -1; }
            break;
        default: why = "not a column form: " + key; return -1;
        }
        cols.push_back(c);
        index[key] = (int)cols.size() - 1;
        return (int)cols.size() - 1;
    }
};

// predicate + aggregate inputs -> ScanColumn filters and ScanAgg products; shared by scans and group scans
template <typename Plan>
bool lower_common(const RowP &pred, const std::vector<RowP> &data, const std::vector<int> &kind, Plan &sp, VCols &col_index, std::string &why) {
    Clause cl;
    if (!to_clause(pred, cl)) {
        std::ostringstream o; o << "predicate is not a conjunction of per-column ranges: ";
        show_row(pred, o);
        why = o.str();
        return false;
    }
    sp.never = cl.never;
    if (!col_index.lower_filters(cl, sp.never)) return false;
    for (size_t j = 0; j < data.size(); j++) {
        ScanAgg ag;
        ag.kind = kind[j];
        if (ag.kind == AGG_FIRST) {
            const int c = col_index(data[j]);
            if (c < 0 || sp.cols[(size_t)c].kind != VC_DIRECT) { if (why.empty()) why = "FoldChoose source is not a table column"; return false; }
            ag.fac.push_back(ScanFactor{c, 0, 1});
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
            const int ci = col_index(f.atom);
            if (ci < 0) return false;
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

// tidy_columns for an aggregate scan: its aggregates (and group key) refer to columns by index
template <typename Plan>
void tidy_scan(Plan &sp, std::vector<KeyStep> *key) {
    std::vector<int> keep;
    for (const ScanAgg &ag : sp.aggs) for (const ScanFactor &f : ag.fac) keep.push_back(f.col);
    if (key) for (const KeyStep &k : *key) if (k.kind == KeyStep::LOAD) keep.push_back(k.col);
    const std::vector<int> map = VCols::tidy_columns(sp.cols, keep);
    for (ScanAgg &ag : sp.aggs) for (ScanFactor &f : ag.fac) f.col = map[(size_t)f.col];
    if (key) for (KeyStep &k : *key) if (k.kind == KeyStep::LOAD) k.col = map[(size_t)k.col];
}

}  // namespace

int64_t eval_scalar(const Scalar &s, const int64_t *agg) {
    switch (s.k) {
    case Scalar::AGG: return agg[s.agg];
    case Scalar::CONST: return s.c;
    default: return apply_bin(s.bin, eval_scalar(*s.l, agg), eval_scalar(*s.r, agg));
    }
}

static void show_col(const ScanColumn &c, size_t k, std::ostringstream &o);
// ProjPlan (vdl_fuse.h): the selection of the program's Partition key, and the atom statements living on it that the
// rest of the program reads.
static void build_projection(const Program &P, Builder &B, FusedPlan &F) {
    ProjPlan &J = F.proj;
    int part = -1;
    for (int id : P.order) if (P.at(id).op == Op::Partition) { if (part < 0) part = id; }
    if (part < 0) { J.why = "no Partition"; return; }
    const Sym &pt = B.sym[(size_t)part];
    if (pt.kind != Sym::PART || pt.sel == 0) { J.why = "the Partition key is not a filtered row expression"; return; }
    J.table = pt.table;
    // candidates: statements whose vector is an atom (or the row ids) on exactly that selection -- or an element-wise expression
    // over such atoms that has the two-accumulator program form of the group keys (emit_key): the scan then evaluates it for the
    // survivors and the executor never sees its operands (Q3: the composite key and the revenue term instead of seven columns)
    struct DryIndex { int operator()(const RowP &) const { return 0; } } dry;
    std::vector<char> cand(P.nodes.size(), 0), is_expr(P.nodes.size(), 0);
    std::vector<int> expr_steps(P.nodes.size(), 0);
    const bool exprs_on = !getenv("VDL_NO_FRONT_EXPR");
    for (int id : P.order) {
        const Sym &s = B.sym[(size_t)id];
        if (s.kind != Sym::ROW || s.table != pt.table || s.sel != pt.sel) continue;
        const Op op = P.at(id).op;
        if (op == Op::Project || op == Op::Shuffle || op == Op::Materialize) continue;      // aliases: their operand is the candidate
        const bool ids = s.e->k == Row::IOTA && s.e->c0 == 0 && s.e->c1 == 1;
        if (ids || (is_atom(s.e) && s.e->k != Row::BIN)) { cand[(size_t)id] = 1; continue; }
        if (exprs_on && s.e->k == Row::BIN && op == Op::Binary) {
            std::vector<KeyStep> trial;
            if (emit_key(s.e, 0, trial, dry) && (int)trial.size() <= kMaxKeySteps) { cand[(size_t)id] = 1; is_expr[(size_t)id] = 1; expr_steps[(size_t)id] = (int)trial.size(); }
        }
    }
    // which of them does the rest of the program read?  (walk back from the outputs, stopping at candidates; when the
    // expressions in use do not fit the descriptor's step pool together, the longest one stops being a candidate)
    std::vector<char> needed, used;
    for (;;) {
        needed.assign(P.nodes.size(), 0); used.assign(P.nodes.size(), 0);
        for (int id : P.outputs) needed[(size_t)id] = 1;
        for (auto it = P.order.rbegin(); it != P.order.rend(); ++it) {
            const Node &n = P.at(*it);
            if (!needed[(size_t)n.id]) continue;
            if (cand[(size_t)n.id]) { used[(size_t)n.id] = 1; continue; }
            for (int opnd : {n.a, n.b, n.c}) if (opnd > 0) needed[(size_t)opnd] = 1;
        }
        int total = 0, nex = 0, longest = -1;
        for (int id : P.order)
            if (used[(size_t)id] && is_expr[(size_t)id]) {
                total += expr_steps[(size_t)id]; nex++;
                if (longest < 0 || expr_steps[(size_t)id] > expr_steps[(size_t)longest]) longest = id;
            }
        if (total > kMaxKeySteps || nex > kMaxProjOuts) { cand[(size_t)longest] = 0; is_expr[(size_t)longest] = 0; continue; }
        // distinct vectors to produce (statements that are the same expression share one; row ids cost none): with more than the
        // take pass writes, the expressions stop being candidates altogether -- the front is then what it was without them
        std::set<std::string> distinct;
        for (int id : P.order)
            if (used[(size_t)id] && B.sym[(size_t)id].e->k != Row::IOTA) distinct.insert(row_key(B.sym[(size_t)id].e));
        if ((int)distinct.size() <= kMaxProjOuts || nex == 0) break;
        for (int id : P.order) if (is_expr[(size_t)id]) { cand[(size_t)id] = 0; is_expr[(size_t)id] = 0; }
    }
    Clause cl;
    if (!to_clause(B.pred_of(pt.sel), cl)) {
        std::ostringstream o; o << "the selection is not a conjunction of per-column ranges / conditions: ";
        show_row(B.pred_of(pt.sel), o);
        J.why = o.str().substr(0, 600);
        return;
    }
    J.never = cl.never;
    VCols vc{J.cols, F, B, J.why, {}};
    if (!vc.lower_filters(cl, J.never)) return;
    for (int id : P.order) {
        if (!used[(size_t)id]) continue;
        const Sym &s = B.sym[(size_t)id];
        int c = -1;
        if (is_expr[(size_t)id]) {
            std::vector<KeyStep> prog;
            if (!emit_key(s.e, 0, prog, vc)) { if (J.why.empty()) J.why = "an expression of the front has an operand without a column form"; return; }
            c = -2 - (int)J.exprs.size();
            J.exprs.push_back(prog);
        } else if (s.e->k != Row::IOTA) { c = vc(s.e); if (c < 0) return; }
        J.nodes.push_back(id); J.node_col.push_back(c);
    }
    if (J.nodes.empty()) { J.why = "nothing downstream reads a column on the Partition's selection"; return; }
    {
        std::set<int> outs;
        for (int nc : J.node_col) if (nc != -1) outs.insert(nc);
        if ((int)outs.size() > kMaxProjOuts) { J.why = "more than " + std::to_string(kMaxProjOuts) + " vectors to produce"; return; }
    }
    // a range check that a lookup through the same index column performs anyway needs no column of its own (the first such
    // lookup then decides about the row and is not deferred to the survivors: run_projection)
    {
        std::vector<int> keep = J.node_col;                       // (what the expressions load stays a column too)
        for (const auto &prog : J.exprs) for (const KeyStep &st : prog) if (st.kind == KeyStep::LOAD) keep.push_back(st.col);
        const std::vector<int> map = VCols::tidy_columns(J.cols, keep);
        for (int &nc : J.node_col) if (nc >= 0) nc = map[(size_t)nc];
        for (auto &prog : J.exprs) for (KeyStep &st : prog) if (st.kind == KeyStep::LOAD) st.col = map[(size_t)st.col];
    }
    if ((int)J.cols.size() > kMaxProjCols) {
        std::ostringstream o;
        o << "more than " << kMaxProjCols << " columns (" << J.cols.size() << "):";
        for (size_t k = 0; k < J.cols.size(); k++) { std::ostringstream one; show_col(J.cols[k], k, one); std::string t = one.str(); o << " |" << t.substr(0, t.size() - 1); }
        J.why = o.str();
        return;
    }
    bool direct = false;
    for (const ScanColumn &c : J.cols) direct |= c.kind == VC_DIRECT && c.name.compare(0, J.table.size() + 1, J.table + ".") == 0;
    if (!direct) { J.why = "no column of the table itself (row count unknown)"; return; }
    J.why.clear();
    J.ok = true;
}

// ---- program rewrites ahead of planning -------------------------------------------------------------------------------------
// "The column of the group's first row", as the reference's compiler writes it (Vlite.hs: an aggregate's representative read
// through the row id the FoldChoose picked; TPC-H Q15 `lineitem_supplier`, Q18 `lineitem_orders`):
//     R = Gather(RangeV 0 .. step 1 over table T, S)      row ids of the selected rows            (or R = the RangeV itself: no filter)
//     D = Scatter(R, .., P)                               ... in group order
//     I = FoldChoose(C, D)                                the first row id of every group
//     G = Gather(X, I)                                    column X of T at that row            (X: a loaded column of T, every slot valid)
// is   G = FoldChoose(C, Scatter(Gather(X, S), .., P)):   the column travels through the same positions and the fold picks the value of
// the same slot (the two Gathers through S hold a value in exactly the same slots: X is a loaded column as long as the RangeV).
// What it buys: the statement no longer reads a column of T by row NUMBER above the GROUP BY -- the column becomes one more vector of
// the filter (a fused front takes it with the survivors), and a sharded run need not have T's rows of other ranks (Q15 then takes the
// front route, vdl_comm.cpp).  The row-id statements stay for whoever else reads them; unread they never run.
void rewrite_program(Program &P) {
    if (getenv("VDL_NO_REWRITE")) return;
    auto alias = [&](int id) { while (id > 0 && (P.at(id).op == Op::Project || P.at(id).op == Op::Shuffle)) id = P.at(id).a; return id; };
    auto table_of = [](const std::string &column) { const size_t dot = column.find('.'); return dot == std::string::npos ? column : column.substr(0, dot); };
    // row ids of a whole table: RangeV 0 step 1 over (an alias of) a Load; "" if not
    auto rowids_of = [&](int id) -> std::string {
        const Node &r = P.at(alias(id));
        if (r.op != Op::RangeV || r.imm0 != 0 || r.imm1 != 1) return "";
        const Node &l = P.at(alias(r.a));
        return l.op == Op::Load ? table_of(l.column) : "";
    };
    const std::vector<int> order = P.order;
    int next_id = 0;
    for (int id : order) next_id = std::max(next_id, id);
    std::vector<int> out;
    out.reserve(order.size() + 8);
    for (int id : order) {
        Node g = P.at(id);
        bool done = false;
        if (g.op == Op::Gather) {
            // (copies, not references: the new statements below grow P.nodes, and what pointed into it would dangle -- found by
            // AddressSanitizer once tools/sanitize ran this function, round 4)
            const Node x = P.at(alias(g.a));
            const Node fc = P.at(alias(g.b));
            if (x.op == Op::Load && x.column.find(".heap") == std::string::npos && fc.op == Op::FoldChoose) {
                const Node sc = P.at(alias(fc.b));
                if (sc.op == Op::Scatter) {
                    const Node r = P.at(alias(sc.a));
                    const std::string t = table_of(x.column);
                    int through = -2;                                  // -1: no filter; >= 0: the selection S
                    if (r.op == Op::Gather && rowids_of(r.a) == t) through = r.b;
                    else if (rowids_of(sc.a) == t) through = -1;
                    if (through != -2 && (size_t)next_id + 2 < ((size_t)1 << 24)) {
                        int src = g.a;
                        if (through >= 0) {
                            Node n1; n1.id = ++next_id; n1.op = Op::Gather; n1.a = g.a; n1.b = through; n1.field = "val"; n1.line = g.line;
                            if (P.nodes.size() <= (size_t)n1.id) P.nodes.resize((size_t)n1.id + 64);
                            P.nodes[(size_t)n1.id] = n1; out.push_back(n1.id); src = n1.id;
                        }
                        // (the length: the old Scatter's size reference when it is anything but the scattered row ids themselves or a range
                        // over them -- otherwise the new source, as long, so that nothing above the Partition hangs off the row ids any more)
                        const Node szn = P.at(alias(sc.b));
                        const bool own_length = alias(sc.b) == alias(sc.a) || (szn.op == Op::RangeV && alias(szn.a) == alias(sc.a));
                        Node n2; n2.id = ++next_id; n2.op = Op::Scatter; n2.a = src; n2.b = own_length ? src : sc.b; n2.c = sc.c; n2.field = "val"; n2.line = g.line;
                        if (P.nodes.size() <= (size_t)n2.id) P.nodes.resize((size_t)n2.id + 64);
                        P.nodes[(size_t)n2.id] = n2; out.push_back(n2.id);
                        g.op = Op::FoldChoose; g.a = fc.a; g.b = n2.id; g.c = -1;
                        P.nodes[(size_t)id] = g;
                        done = true;
                    }
                }
            }
        }
        (void)done;
        out.push_back(id);
    }
    P.order = out;
}

FusedPlan fuse_program(const Program &P) {
    FusedPlan F;
    Builder B(P);
    for (int id : P.order) {
        B.sym[(size_t)id] = B.visit(P.at(id));
        const Sym &s = B.sym[(size_t)id];
        if (s.kind == Sym::ROW && s.sel && !B.witness.count(s.sel)) B.witness[s.sel] = id;
    }
    F.filters = B.filters;
    if (P.outputs.empty()) { F.why_not = "program has no MaterializeCompact output"; return F; }
    build_projection(P, B, F);
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
        VCols vc{sp.cols, F, B, F.why_not, {}};
        if (!lower_common(B.pred_of(ps.sel), ps.data, ps.kind, sp, vc, F.why_not)) return F;
        tidy_scan(sp, (std::vector<KeyStep> *)nullptr);
        bool direct = false;
        for (const ScanColumn &c : sp.cols) direct |= c.kind == VC_DIRECT && c.name.compare(0, sp.table.size() + 1, sp.table + ".") == 0;
        if (!direct) { F.why_not = "scan touches no column of its table (row count unknown)"; return F; }
        if ((int)sp.cols.size() > kMaxJoinScanCols) { F.why_not = "scan needs more than 12 columns (table columns, lookups, conditions)"; return F; }
        if ((int)sp.aggs.size() > kMaxScanAggs) { F.why_not = "scan has more than 8 aggregates"; return F; }
        F.scans.push_back(sp);
    }
    for (auto &pg : B.groups) {
        GroupScanPlan gp;
        gp.table = pg.table; gp.pmin = pg.pmin; gp.pcount = pg.pcount;
        VCols col_index{gp.cols, F, B, F.why_not, {}};
        if (!lower_common(B.pred_of(pg.sel), pg.data, pg.kind, gp, col_index, F.why_not)) return F;
        if (!emit_key(pg.key, 0, gp.key, col_index)) {
            if (F.why_not.empty()) {
                std::ostringstream o; o << "group key is not a chain of single-column terms: ";
                show_row(pg.key, o);
                F.why_not = o.str();
            }
            return F;
        }
        {
            bool direct = false;
            for (const ScanColumn &c : gp.cols) direct |= c.kind == VC_DIRECT && c.name.compare(0, gp.table.size() + 1, gp.table + ".") == 0;
            if (!direct) { F.why_not = "grouped scan touches no column of its table (row count unknown)"; return F; }
        }
        tidy_scan(gp, &gp.key);
        if ((int)gp.key.size() > kMaxKeySteps) { F.why_not = "group key program too long"; return F; }
        if ((int)gp.cols.size() > kMaxJoinScanCols) { F.why_not = "grouped scan needs more than 12 columns (table columns, lookups, conditions)"; return F; }
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
    F.why_not.clear();
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

static void show_col(const ScanColumn &c, size_t k, std::ostringstream &o) {
    o << "  col " << k << " ";
    switch (c.kind) {
    case VC_DIRECT: o << c.name; break;
    case VC_GATHER: o << c.name << "[col" << c.idx << "]"; break;
    case VC_BITS: o << "prelude" << c.prelude << ".bit[col" << c.idx << "]"; break;
    case VC_LUT: o << "prelude" << c.prelude << ".lut[col" << c.idx << "]"; break;
    case VC_INRANGE: o << "inrange(col" << c.idx << ", rows of " << c.name << ")"; break;
    case VC_ROWID: o << "row id"; break;
    case VC_FORM:
        o << "cond(";
        for (size_t i = 0; i < c.form.size(); i++) {
            const FormStep &f = c.form[i];
            if (i) o << " ";
            if (f.op == FormStep::LEAF) {
                o << "col" << f.col << ":[";
                if (f.lo == INT64_MIN) o << "-inf"; else o << f.lo;
                o << ",";
                if (f.hi == INT64_MAX) o << "+inf"; else o << f.hi;
                o << "]";
            } else o << (f.op == FormStep::AND ? "and" : f.op == FormStep::OR ? "or" : f.op == FormStep::NOT ? "not" : f.op == FormStep::TRUE_ ? "true" : "false");
        }
        o << ")";
        break;
    default: o << "col" << c.idx << " - col" << c.idx2; break;
    }
    if (c.lo != INT64_MIN || c.hi != INT64_MAX) {
        o << " in [";
        if (c.lo == INT64_MIN) o << "-inf"; else o << c.lo;
        o << ",";
        if (c.hi == INT64_MAX) o << "+inf"; else o << c.hi;
        o << "]";
    }
    o << "\n";
}

static void show_prelude(const FusedPlan &F, std::ostringstream &o) {
    for (size_t k = 0; k < F.prelude.size(); k++) {
        const PreludeItem &it = F.prelude[k];
        if (it.kind == PreludeItem::LIKE_LUT) { o << "prelude " << k << ": LIKE '" << it.pattern << "' over every offset of " << it.heap << "\n"; continue; }
        if (it.kind == PreludeItem::SEMI_BITMAP) {
            o << "prelude " << k << ": semi-join set over the rows of " << it.bits_of.substr(0, it.bits_of.find('.')) << ": one scan of " << it.table << (it.never ? " [never]" : "")
              << " sets bit col" << it.index_col;
            if (it.modulus) o << " (emitted mod " << it.modulus << ": exact while the table has no more rows)";
            o << "\n";
            for (size_t c = 0; c < it.cols.size(); c++) show_col(it.cols[c], c, o);
            continue;
        }
        o << "prelude " << k << ": bitmap of the dimension-side selection held by statement " << it.witness;
        if (!it.scan) { o << " (per-operator executor)\n"; continue; }
        o << ": one scan of " << it.table << (it.never ? " [never]" : "") << "\n";
        for (size_t c = 0; c < it.cols.size(); c++) show_col(it.cols[c], c, o);
    }
}

std::string describe_fused(const FusedPlan &F) {
    std::ostringstream o;
    if (!F.ok) {
        o << "not fused: " << F.why_not << "\n";
        if (F.proj.ok) {
            show_prelude(F, o);
            o << "fused front: one scan of " << F.proj.table << (F.proj.never ? " [never]" : "") << " hands these statements to the per-operator executor as sparse vectors:";
            for (size_t k = 0; k < F.proj.nodes.size(); k++)
                o << " Id " << F.proj.nodes[k] << (F.proj.node_col[k] == -1 ? "=rowid" : F.proj.node_col[k] < -1 ? "=expr" + std::to_string(-2 - F.proj.node_col[k]) + "(" +
                     std::to_string(F.proj.exprs[(size_t)(-2 - F.proj.node_col[k])].size()) + " steps)" : "=col" + std::to_string(F.proj.node_col[k]));
            o << "\n";
            for (size_t c = 0; c < F.proj.cols.size(); c++) show_col(F.proj.cols[c], c, o);
        } else if (!F.proj.why.empty()) {
            o << "no fused front: " << F.proj.why << "\n";
        }
        return o.str();
    }
    show_prelude(F, o);
    for (size_t i = 0; i < F.scans.size(); i++) {
        const ScanPlan &sp = F.scans[i];
        o << "scan " << i << " table=" << sp.table << (sp.never ? " [never]" : "") << "\n";
        for (size_t c = 0; c < sp.cols.size(); c++) show_col(sp.cols[c], c, o);
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
        for (size_t c = 0; c < gp.cols.size(); c++) show_col(gp.cols[c], c, o);
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

// vdl_exchange.cpp -- sharded Partition: which vectors travel (analysis) and the three C-ABI calls around the
// caller's all-to-all (vdl_exchange_begin / _pack / _finish); see vdl_partition.hip "Row exchange".
#include "vdl_genexec.h"

namespace {

// ------------------------------------------------------------------------------------------------
// sharded Partition: exchange analysis (which vectors travel) -- see vdl_partition.hip "Row exchange"
// ------------------------------------------------------------------------------------------------
struct ExchangeSpec {
    bool ok = false;
    std::string why;
    int part = 0, key = 0;             // Partition statement, its (resolved) data operand
    std::vector<int> sources;          // resolved source statements of the Scatters that use the partition; [0] = key
    int64_t pmin = 0, pcount = 0;
    std::vector<int> folds;            // global folds over rows of the sharded table that the tail reads beside the Partition
};

int resolve_alias(const Program &P, int id) {
    while (P.at(id).op == Op::Project || P.at(id).op == Op::Shuffle) id = P.at(id).a;
    return id;
}

// Class of every statement below `roots` when `table` is split by rows over the ranks: R = replicated (same on every
// rank), V = one value per row of the shard, N = (local) row numbers of the shard, as the filter idiom
// Gather(x, FoldSelect(RangeV 0 step 1, cond)) of Vlite.hs produces them.  false + why: something is not row-local.
enum : char { R = 0, V = 1, N = 2 };
bool classify_rows(const Program &P, const std::vector<int> &roots, const std::string &table, const char *boundary,
                   std::vector<char> &cls, std::string &why) {
    std::vector<char> below(P.nodes.size(), 0);
    cls.assign(P.nodes.size(), R);
    std::vector<int> stack(roots.begin(), roots.end());
    while (!stack.empty()) {
        const int id = stack.back(); stack.pop_back();
        if (below[(size_t)id]) continue;
        below[(size_t)id] = 1;
        const Node &n = P.at(id);
        for (int opnd : {n.a, n.b, n.c}) if (opnd > 0) stack.push_back(opnd);
    }
    for (int id : P.order) {
        if (!below[(size_t)id]) continue;
        const Node &n = P.at(id);
        auto C = [&](int o) { return o > 0 ? cls[(size_t)o] : (char)R; };
        const std::string at = " (statement " + std::to_string(id) + ")";
        char &out = cls[(size_t)id];
        switch (n.op) {
        case Op::Load: out = n.column.compare(0, table.size() + 1, table + ".") == 0 ? V : R; break;
        case Op::RangeC: break;
        case Op::Project: case Op::Shuffle: case Op::Materialize: out = C(n.a); break;
        case Op::Like:
            if (C(n.b) != R) { why = "Like over a sharded string heap" + at; return false; }
            if (C(n.a) == N) { why = "Like on row numbers" + at; return false; }
            out = C(n.a);
            break;
        case Op::Binary:
            if (C(n.a) == N || C(n.b) == N) { why = "arithmetic on row numbers of the sharded table, which are rank-local" + at; return false; }
            out = (C(n.a) == V || C(n.b) == V) ? V : R;
            break;
        case Op::RangeV:
            if (C(n.a) == R) break;
            if (n.imm1 == 0) out = V;
            else if (n.imm0 == 0 && n.imm1 == 1) out = N;
            else { why = "a strided range over the sharded table is rank-local" + at; return false; }
            break;
        case Op::FoldSelect: {
            if (C(n.a) == R && C(n.b) == R) break;
            const Node &ctl = P.at(resolve_alias(P, n.a));
            if (!(ctl.op == Op::RangeV && C(n.a) == N)) { why = "FoldSelect over runs of the sharded table" + at; return false; }
            out = N;
            break;
        }
        case Op::Gather:
            if (C(n.a) == R && C(n.b) != N) out = C(n.b);                 // replicated data by FK / replicated positions
            else if (C(n.a) != R && C(n.b) == N) out = C(n.a);            // shard data by shard row numbers
            else { why = "Gather mixes replicated and rank-local positions" + at; return false; }
            break;
        default:
            if (C(n.a) != R || C(n.b) != R || C(n.c) != R) { why = std::string(op_name(n.op, n.bin)) + " over the sharded table below " + boundary + at; return false; }
        }
    }
    return true;
}

// `table`: name of the row-sharded table ("" = trust the caller).  With a table name the statements
// below the scatters are checked to be row-local over that table: its columns may pass through
// element-wise operators, constants and Gathers *from* replicated vectors only.
// allow_folds: the caller merges global fold records across the ranks (vdl_run_sharded); then a global Fold over row-local data
// of the sharded table may stand beside the Partition and what is above it may read its (merged) result.
enum : char { TR = 0, TG = 1, TI = 2, TS = 3 };     // classes above the cut of analyse_exchange (see there)
// gather_all (the second cut of the "chain" route, below): the rows that reach the FIRST Partition over the sharded table go to EVERY rank,
// rank after rank = row order, and the tail runs everywhere on all of them: further Partitions may stand above the cut and the tail
// need not treat every group by itself.  tail_class: the classes of the statements above the cut (TR / TG / TI / TS, below).
ExchangeSpec analyse_exchange(const Program &P, const std::string &table = std::string(), bool allow_folds = false, bool gather_all = false,
                              std::vector<char> *tail_class = nullptr) {
    ExchangeSpec x;
    std::vector<char> needed(P.nodes.size(), 0);
    for (int id : P.outputs) needed[(size_t)id] = 1;
    for (auto it = P.order.rbegin(); it != P.order.rend(); ++it) {
        const Node &n = P.at(*it);
        if (!needed[(size_t)n.id]) continue;
        for (int opnd : {n.a, n.b, n.c}) if (opnd > 0) needed[(size_t)opnd] = 1;
    }
    std::vector<char> reads_table(P.nodes.size(), 0);
    for (int id : P.order) {
        const Node &n = P.at(id);
        if (n.op == Op::Load) { reads_table[(size_t)id] = !table.empty() && n.column.compare(0, table.size() + 1, table + ".") == 0; continue; }
        for (int opnd : {n.a, n.b, n.c}) if (opnd > 0 && reads_table[(size_t)opnd]) reads_table[(size_t)id] = 1;
    }
    for (int id : P.order) {
        if (!needed[(size_t)id] || P.at(id).op != Op::Partition) continue;
        if (gather_all) {
            if (!x.part && reads_table[(size_t)id]) x.part = id;
            continue;
        }
        if (x.part) { x.why = "more than one Partition"; return x; }
        x.part = id;
    }
    if (!x.part) { x.why = "no Partition in the program"; return x; }
    const Node &pn = P.at(x.part);
    const Node &piv = P.at(resolve_alias(P, pn.b));
    if (piv.op != Op::RangeC || piv.imm2 != 1 || piv.imm1 <= 0) { x.why = "pivots are not a RangeC with step 1"; return x; }
    x.pmin = piv.imm0; x.pcount = piv.imm1;
    x.key = resolve_alias(P, pn.a);
    x.sources.push_back(x.key);
    std::vector<char> is_cut(P.nodes.size(), 0);
    for (int id : P.order) {
        const Node &n = P.at(id);
        if (!needed[(size_t)id]) continue;
        bool uses = false;
        for (int opnd : {n.a, n.b, n.c}) uses |= opnd > 0 && resolve_alias(P, opnd) == x.part && !(n.op == Op::Project || n.op == Op::Shuffle);
        if (!uses) continue;
        if (n.op != Op::Scatter || resolve_alias(P, n.c) != x.part || resolve_alias(P, n.a) == x.part || resolve_alias(P, n.b) == x.part) {
            x.why = "the Partition result is used other than as Scatter positions (statement " + std::to_string(id) + ")";
            return x;
        }
        is_cut[(size_t)id] = 1;
        const int src = resolve_alias(P, n.a);
        if (std::find(x.sources.begin(), x.sources.end(), src) == x.sources.end()) x.sources.push_back(src);
    }
    if ((int)x.sources.size() - 1 > kMaxExSources) { x.why = "too many scattered vectors"; return x; }
    // everything above the scatters must be derived from them alone
    std::vector<char> seen(P.nodes.size(), 0);
    std::vector<int> stack(P.outputs.begin(), P.outputs.end());
    while (!stack.empty()) {
        const int id = stack.back(); stack.pop_back();
        if (seen[(size_t)id]) continue;
        seen[(size_t)id] = 1;
        const Node &n = P.at(id);
        if (is_cut[(size_t)id]) continue;
        if (allow_folds && !table.empty() && (n.op == Op::FoldSum || n.op == Op::FoldMin || n.op == Op::FoldMax || n.op == Op::FoldCount)) {
            const Node &ctl = P.at(resolve_alias(P, n.a));
            if (ctl.op == Op::RangeV && ctl.imm1 == 0) {          // one run over everything: a candidate (kept if its data is row-local)
                if (std::find(x.folds.begin(), x.folds.end(), id) == x.folds.end()) x.folds.push_back(id);
                continue;
            }
        }
        if (n.op == Op::Load) {
            // columns of the other (replicated) tables are there on every rank: the tail may gather from them by values
            // that travelled (Q10 prints customer columns through the group's FK value)
            if (!table.empty() && n.column.compare(0, table.size() + 1, table + ".") != 0) continue;
            x.why = "output depends on column " + n.column + " other than through the partition";
            return x;
        }
        if (id == x.part) { x.why = "Partition reachable past the scatters"; return x; }
        for (int opnd : {n.a, n.b, n.c}) if (opnd > 0) stack.push_back(opnd);
    }
    // size references of the scatters are evaluated after the exchange: they must hang off the travelling vectors
    std::vector<char> is_src(P.nodes.size(), 0);
    for (int id : x.sources) is_src[(size_t)id] = 1;
    std::fill(seen.begin(), seen.end(), 0);
    stack.clear();
    for (int id : P.order) if (is_cut[(size_t)id]) stack.push_back(P.at(id).b);
    while (!stack.empty()) {
        const int id = stack.back(); stack.pop_back();
        if (seen[(size_t)id] || is_src[(size_t)id]) continue;
        seen[(size_t)id] = 1;
        const Node &n = P.at(id);
        if (n.op == Op::Load) { x.why = "a Scatter size reference depends on column " + n.column + " directly"; return x; }
        for (int opnd : {n.a, n.b, n.c}) if (opnd > 0) stack.push_back(opnd);
    }
    if (!table.empty()) {
        std::vector<char> cls;
        std::vector<int> roots = x.sources;
        for (int id : x.folds) { roots.push_back(P.at(id).a); roots.push_back(P.at(id).b); }
        if (!classify_rows(P, roots, table, "the Partition", cls, x.why)) return x;
        for (int id : x.sources) {
            if (cls[(size_t)id] == N) { x.why = "statement " + std::to_string(id) + " feeds the Partition with rank-local row numbers"; return x; }
            if (cls[(size_t)id] != V) { x.why = "statement " + std::to_string(id) + " feeds the Partition but does not depend on table " + table; return x; }
        }
        // a candidate fold over replicated data is the same on every rank: no cut (nothing below it reads the sharded table)
        std::vector<int> kept;
        for (int id : x.folds) {
            const char cd = cls[(size_t)P.at(id).b], cc = cls[(size_t)P.at(id).a];
            if (cd == N) { x.why = "statement " + std::to_string(id) + " folds rank-local row numbers"; return x; }
            if (cd == V || cc == V) kept.push_back(id);
        }
        std::sort(kept.begin(), kept.end());
        x.folds = kept;
    } else {
        x.folds.clear();
    }
    if (gather_all) { x.folds.clear(); x.ok = true; return x; }
    // What stands ABOVE the scatters runs on every rank over the groups of ITS key range, and the ranks' outputs are concatenated: that is
    // the unsharded answer only if the tail treats every group by itself.  (Round 4: TPC-H Q20 was accepted although its tail feeds a
    // semi-join set over suppliers from the groups -- a supplier whose qualifying groups lie on two ranks came out twice; it went unseen
    // while the tests' keys filled so little of their declared domain that the even cut sent every row to rank 0.)  Classes above the cut:
    // R replicated / scalar, G one slot per received row, I the slots' own ids, S positions of a selection of slots.
    {
        std::vector<char> above(P.nodes.size(), 0), tc(P.nodes.size(), TR);
        std::vector<int> st(P.outputs.begin(), P.outputs.end());
        while (!st.empty()) {
            const int id = st.back(); st.pop_back();
            if (above[(size_t)id]) continue;
            above[(size_t)id] = 1;
            if (is_cut[(size_t)id] || std::find(x.folds.begin(), x.folds.end(), id) != x.folds.end()) continue;
            const Node &n = P.at(id);
            for (int opnd : {n.a, n.b, n.c}) if (opnd > 0) st.push_back(opnd);
        }
        auto is_key_scatter = [&](int id) { id = resolve_alias(P, id); return id > 0 && is_cut[(size_t)id] && resolve_alias(P, P.at(id).a) == x.key; };
        for (int id : P.order) {
            if (!above[(size_t)id]) continue;
            const Node &n = P.at(id);
            auto C = [&](int o) { return o > 0 ? tc[(size_t)o] : (char)TR; };
            const std::string at = " (statement " + std::to_string(id) + ": the tail above the Partition does not treat every group by itself)";
            char &out = tc[(size_t)id];
            if (is_cut[(size_t)id]) { out = TG; continue; }
            if (std::find(x.folds.begin(), x.folds.end(), id) != x.folds.end()) { out = TR; continue; }      // a merged global fold: a scalar
            const bool all_r = C(n.a) == TR && C(n.b) == TR && C(n.c) == TR;
            switch (n.op) {
            case Op::Load: case Op::RangeC: out = TR; break;
            case Op::Project: case Op::Shuffle: case Op::Materialize: out = C(n.a); break;
            case Op::Like:
                if (C(n.b) != TR || C(n.a) == TI || C(n.a) == TS) { x.why = "Like over rank-local values" + at; return x; }
                out = C(n.a);
                break;
            case Op::Binary:
                if (C(n.a) >= TI || C(n.b) >= TI) { x.why = "arithmetic on rank-local slot numbers" + at; return x; }
                out = (C(n.a) == TG || C(n.b) == TG) ? TG : TR;
                break;
            case Op::RangeV:
                if (C(n.a) == TR) { out = TR; break; }
                if (C(n.a) != TG) { x.why = "a range over rank-local positions" + at; return x; }
                if (n.imm1 == 0) out = TG;
                else if (n.imm0 == 0 && n.imm1 == 1) out = TI;
                else { x.why = "a strided range over the groups" + at; return x; }
                break;
            case Op::FoldSelect:
                if (all_r) { out = TR; break; }
                if (C(n.a) != TI || C(n.b) != TG) { x.why = "FoldSelect over runs of groups" + at; return x; }
                out = TS;
                break;
            case Op::FoldSum: case Op::FoldMin: case Op::FoldMax: case Op::FoldCount: case Op::FoldChoose:
                if (all_r) { out = TR; break; }
                // the runs must be the Partition key's own (one run = one group): anything else folds ACROSS groups
                if (!is_key_scatter(n.a) || C(n.b) >= TI) { x.why = std::string(op_name(n.op, n.bin)) + " over runs that are not the Partition key's" + at; return x; }
                out = TG;
                break;
            case Op::Gather:
                if (all_r) out = TR;
                else if (C(n.a) == TR && C(n.b) == TG) out = TG;             // a replicated table looked up by a group's value
                else if (C(n.a) == TG && (C(n.b) == TS || C(n.b) == TI)) out = TG;   // the filter idiom: groups picked by their own slot numbers
                else { x.why = "Gather across groups" + at; return x; }
                break;
            case Op::Scatter:
                if (all_r) { out = TR; break; }
                // (a Scatter back to the slots' own ids is the source restricted to a selection; any other one moves values between groups
                // or into a table the ranks do not share: Q20's semi-join set)
                if (C(n.a) <= TG && C(n.b) == TG && C(n.c) == TI) out = TG;
                else { x.why = "Scatter by positions other than the Partition's" + at; return x; }
                break;
            default:
                if (!all_r) { x.why = std::string(op_name(n.op, n.bin)) + " above the Partition" + at; return x; }
                out = TR;
            }
        }
        if (tail_class) *tail_class = tc;
    }
    x.ok = true;
    return x;
}


// ------------------------------------------------------------------------------------------------
// General plans whose outputs hang off GLOBAL folds over the sharded table (join + ungrouped aggregate: Q14, Q19):
// each rank folds its rows, the fold results travel as mergeable words (vdl_plan_partial_spec / vdl_run_local), and the
// statements above the folds run on the merged scalars (vdl_finalize).  Same three calls as for fused plans.
// ------------------------------------------------------------------------------------------------
struct FoldCut {
    bool ok = false;
    std::string why;
    std::vector<int> folds;            // global Fold{Sum,Min,Max,Count} statements over row-local data, program order
};

FoldCut analyse_folds(const Program &P, const std::string &table) {
    FoldCut x;
    if (table.empty()) { x.why = "no row-sharded table named (vdl_plan_set_sharded_table)"; return x; }
    std::vector<char> needed(P.nodes.size(), 0);
    for (int id : P.outputs) needed[(size_t)id] = 1;
    for (auto it = P.order.rbegin(); it != P.order.rend(); ++it) {
        const Node &n = P.at(*it);
        if (!needed[(size_t)n.id]) continue;
        for (int opnd : {n.a, n.b, n.c}) if (opnd > 0) needed[(size_t)opnd] = 1;
    }
    // dep: reads the sharded table not through a cut; via: derived from a cut (a merged scalar)
    std::vector<char> dep(P.nodes.size(), 0), via(P.nodes.size(), 0);
    for (int id : P.order) {
        if (!needed[(size_t)id]) continue;
        const Node &n = P.at(id);
        const std::string at = " (statement " + std::to_string(id) + ")";
        if (n.op == Op::Load) { dep[(size_t)id] = n.column.compare(0, table.size() + 1, table + ".") == 0; continue; }
        bool d = false, v = false;
        for (int opnd : {n.a, n.b, n.c}) if (opnd > 0) { d |= dep[(size_t)opnd] != 0; v |= via[(size_t)opnd] != 0; }
        const bool fold = n.op == Op::FoldSum || n.op == Op::FoldMin || n.op == Op::FoldMax || n.op == Op::FoldCount;
        if (fold && d && !v) {
            const Node &ctl = P.at(resolve_alias(P, n.a));
            if (ctl.op == Op::RangeV && ctl.imm1 == 0) {          // one run over everything: a global fold
                x.folds.push_back(id);
                via[(size_t)id] = 1;
                continue;
            }
        }
        if (v) {
            // above the folds only scalars: element-wise operators with constants or other merged scalars
            if (d) { x.why = std::string(op_name(n.op, n.bin)) + " combines a global fold result with rows of " + table + at; return x; }
            const bool scalar_op = n.op == Op::Binary || n.op == Op::RangeV || n.op == Op::Project || n.op == Op::Shuffle || n.op == Op::Materialize;
            if (!scalar_op) { x.why = std::string(op_name(n.op, n.bin)) + " uses a global fold result as a vector" + at; return x; }
            if (n.op == Op::Binary)
                for (int opnd : {n.a, n.b})
                    if (!via[(size_t)opnd]) {
                        const Op o = P.at(resolve_alias(P, opnd)).op;
                        if (o != Op::RangeV && o != Op::RangeC) { x.why = "a global fold result meets a stored vector" + at; return x; }
                    }
            via[(size_t)id] = 1;
            continue;
        }
        dep[(size_t)id] = d;
    }
    if (x.folds.empty()) { x.why = "no global fold over table " + table; return x; }
    for (int id : P.outputs)
        if (dep[(size_t)id]) { x.why = "output " + std::to_string(id) + " depends on rows of " + table + " other than through a global fold"; return x; }
    std::vector<int> roots;
    for (int id : x.folds) { roots.push_back(P.at(id).a); roots.push_back(P.at(id).b); }
    std::vector<char> cls;
    if (!classify_rows(P, roots, table, "the global folds", cls, x.why)) return x;
    for (int id : x.folds)
        if (cls[(size_t)P.at(id).b] == N) { x.why = "statement " + std::to_string(id) + " folds rank-local row numbers"; return x; }
    x.ok = true;
    return x;
}

int fold_reduce_kind(Op op) { return op == Op::FoldMin ? 1 : op == Op::FoldMax ? 2 : 0; }       // count merges as a sum

// ------------------------------------------------------------------------------------------------
// The "chain" route (TPC-H Q18, /root/reference/tests/tpch10noorder/18.sql.mplan): a GROUP BY over ALL rows of the sharded table
// whose groups only feed POSITION SETS -- Scatter(constant, size, a value of the group): the semi-join set of `o_orderkey in (select
// l_orderkey .. group by l_orderkey having sum(l_quantity) > 300)`, Vlite.hs:1212-1222 -- and a rest that reads the sets and the table
// a second time.  Three mechanisms the other routes already have, one after the other:
//   stage 1  the rows travel to the owners of their key range (the exchange route's cut, analyse_exchange with the sets' operands as
//            its outputs: the tail up to them must treat every group by itself); every owner runs the GROUP BY on complete groups and
//            packs the positions its groups put into every set;
//   merge    the packed positions are all-gathered (a set of constants is the union of its positions, whoever found them) and every
//            rank builds the same set vectors, as long as the unsharded Scatter would have been;
//   stage 2  the rest of the program with the sets in place: what it computes per row of the table runs on each rank's OWN rows (the
//            shard it was given, not the exchanged rows: rank after rank = row order), the rows that reach the next Partition are
//            all-gathered and the tail runs on every rank (analyse_exchange, gather_all) -- every rank ends with the whole answer.
// The scan of the table, the GROUP BY over all of it and the second scan scale with the ranks; the tail over the survivors does not.
// ------------------------------------------------------------------------------------------------

// Gather(X, Gather(FoldChoose(C, Scatter(row ids [of a selection S], .., P)), sel)) -- a column of the group's first row, looked up by row
// NUMBER for the groups `sel` keeps, which a rank of a sharded run cannot serve -- is Gather(FoldChoose(C, Scatter(X [on S], .., P)), sel):
// the column travels through the fold like any other value of the group.  (rewrite_program, vdl_fuse.cpp, does the same for the form
// without `sel`; this one is kept out of unsharded runs, where it would send a whole column through the Partition for the sake of a few
// groups.)  The new Scatter takes its length from the vector it scatters, so that nothing above the cut hangs off the row ids.
void rewrite_chain_lookups(Program &P) {
    auto alias = [&](int id) { return id > 0 ? resolve_alias(P, id) : id; };
    auto table_of = [](const std::string &column) { const size_t dot = column.find('.'); return dot == std::string::npos ? column : column.substr(0, dot); };
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
    auto add = [&](Node n) {
        n.id = ++next_id; n.field = "val";
        if (P.nodes.size() <= (size_t)n.id) P.nodes.resize((size_t)n.id + 64);
        P.nodes[(size_t)n.id] = n;
        out.push_back(n.id);
        return n.id;
    };
    for (int id : order) {
        Node g = P.at(id);
        if (g.op == Op::Gather && (size_t)next_id + 4 < ((size_t)1 << 24)) {
            const Node x = P.at(alias(g.a)), g2 = P.at(alias(g.b));
            if (x.op == Op::Load && x.column.find(".heap") == std::string::npos && g2.op == Op::Gather) {
                const Node fc = P.at(alias(g2.a));
                if (fc.op == Op::FoldChoose) {
                    const Node sc = P.at(alias(fc.b));
                    const Node szn = sc.op == Op::Scatter ? P.at(alias(sc.b)) : Node();
                    if (sc.op == Op::Scatter && (alias(sc.b) == alias(sc.a) || (szn.op == Op::RangeV && alias(szn.a) == alias(sc.a)))) {
                        const Node r = P.at(alias(sc.a));
                        const std::string t = table_of(x.column);
                        int through = -2;                                  // -1: no filter; >= 0: the selection S
                        if (r.op == Op::Gather && rowids_of(r.a) == t) through = r.b;
                        else if (rowids_of(sc.a) == t) through = -1;
                        if (through != -2) {
                            int src = g.a;
                            if (through >= 0) { Node n1; n1.op = Op::Gather; n1.a = g.a; n1.b = through; n1.line = g.line; src = add(n1); }
                            Node n2; n2.op = Op::Scatter; n2.a = src; n2.b = src; n2.c = sc.c; n2.line = g.line;
                            const int scattered = add(n2);
                            Node n3; n3.op = Op::FoldChoose; n3.a = fc.a; n3.b = scattered; n3.line = g.line;
                            g.a = add(n3); g.b = g2.b;
                            P.nodes[(size_t)id] = g;
                        }
                    }
                }
            }
        }
        out.push_back(id);
    }
    P.order = out;
}

struct ChainSpec {
    bool ok = false;
    std::string why;
    ChainPlan plan;
};

// the program with every set statement standing for a vector all ranks share (stage 2's view)
Program with_sets_given(const Program &P, const std::vector<int> &sets) {
    Program B = P;
    for (int id : sets) {
        Node n; n.id = id; n.op = Op::Load; n.column = "(position set " + std::to_string(id) + ")"; n.field = P.at(id).field; n.line = P.at(id).line;
        B.nodes[(size_t)id] = n;
    }
    return B;
}

ChainSpec analyse_chain(const Program &P0, const std::string &table) {
    ChainSpec ch;
    if (table.empty()) { ch.why = "no row-sharded table named (vdl_plan_set_sharded_table)"; return ch; }
    ChainPlan &cp = ch.plan;
    cp.prog = P0;
    rewrite_chain_lookups(cp.prog);
    const Program &P = cp.prog;
    std::vector<char> needed(P.nodes.size(), 0), reads_table(P.nodes.size(), 0);
    for (int id : P.outputs) needed[(size_t)id] = 1;
    for (auto it = P.order.rbegin(); it != P.order.rend(); ++it) {
        const Node &n = P.at(*it);
        if (!needed[(size_t)n.id]) continue;
        for (int opnd : {n.a, n.b, n.c}) if (opnd > 0) needed[(size_t)opnd] = 1;
    }
    for (int id : P.order) {
        const Node &n = P.at(id);
        if (n.op == Op::Load) { reads_table[(size_t)id] = n.column.compare(0, table.size() + 1, table + ".") == 0; continue; }
        for (int opnd : {n.a, n.b, n.c}) if (opnd > 0 && reads_table[(size_t)opnd]) reads_table[(size_t)id] = 1;
    }
    // raw: depends on the first Partition over the table other than through a position set
    int first = 0;
    std::vector<char> raw(P.nodes.size(), 0);
    for (int id : P.order) {
        if (!needed[(size_t)id]) continue;
        const Node &n = P.at(id);
        if (n.op == Op::Partition && !first && reads_table[(size_t)id]) { first = id; raw[(size_t)id] = 1; continue; }
        bool r = false;
        for (int opnd : {n.a, n.b, n.c}) r |= opnd > 0 && raw[(size_t)opnd] != 0;
        if (!r) continue;
        if (n.op == Op::Scatter && n.c > 0 && raw[(size_t)n.c] && resolve_alias(P, n.c) != first) {
            const Node &v = P.at(resolve_alias(P, n.a));
            if (v.op == Op::RangeV && v.imm1 == 0) {
                cp.sets.push_back(id);
                cp.constant.push_back(v.imm0);
                for (int opnd : {n.a, n.b, n.c}) cp.targets.push_back(opnd);
                continue;                                      // (what reads the set does not read the groups)
            }
        }
        raw[(size_t)id] = 1;
    }
    if (!first) { ch.why = "no Partition over table " + table; return ch; }
    if (cp.sets.empty()) { ch.why = "the first Partition over table " + table + " (statement " + std::to_string(first) + ") feeds no position set"; return ch; }
    for (int id : P.outputs)
        if (raw[(size_t)id]) { ch.why = "output " + std::to_string(id) + " reads the groups of statement " + std::to_string(first) + " other than through a position set"; return ch; }
    // stage 1: the exchange route's analysis with the sets' operands for outputs
    {
        Program A = P;
        A.outputs = cp.targets;
        std::sort(A.outputs.begin(), A.outputs.end());
        A.outputs.erase(std::unique(A.outputs.begin(), A.outputs.end()), A.outputs.end());
        std::vector<char> tc;
        const ExchangeSpec x = analyse_exchange(A, table, false, false, &tc);
        if (!x.ok) { ch.why = "up to its position sets: " + x.why; return ch; }
        if (x.part != first) { ch.why = "the position sets hang off another Partition than the first over table " + table; return ch; }
        for (size_t k = 0; k < cp.sets.size(); k++) {
            const int a = cp.targets[3 * k], b = cp.targets[3 * k + 1], c = cp.targets[3 * k + 2];
            const std::string at = " (statement " + std::to_string(cp.sets[k]) + ")";
            if (tc[(size_t)c] != TG) { ch.why = "the positions of a set are not values of the groups" + at; return ch; }
            if (tc[(size_t)a] != TG && tc[(size_t)a] != TR) { ch.why = "the constant of a set is spread over rank-local slots" + at; return ch; }
            if (tc[(size_t)b] == TS) { ch.why = "the length of a set is that of a selection of groups" + at; return ch; }
            cp.size_replicated.push_back(tc[(size_t)b] == TR);
        }
    }
    // stage 2: the rest, the sets given
    {
        const Program B = with_sets_given(P, cp.sets);
        std::vector<char> need2(B.nodes.size(), 0);
        for (int id : B.outputs) need2[(size_t)id] = 1;
        bool reads = false;
        for (auto it = B.order.rbegin(); it != B.order.rend(); ++it) {
            const Node &n = B.at(*it);
            if (!need2[(size_t)n.id]) continue;
            if (n.op == Op::Load && n.column.compare(0, table.size() + 1, table + ".") == 0) reads = true;
            for (int opnd : {n.a, n.b, n.c}) if (opnd > 0) need2[(size_t)opnd] = 1;
        }
        cp.second_cut = reads;
        if (reads) {
            const ExchangeSpec x = analyse_exchange(B, table, false, true);
            if (!x.ok) { ch.why = "above its position sets: " + (x.part ? x.why : "the rest reads table " + table + " without a Partition to gather its rows at"); return ch; }
        }
    }
    ch.ok = true;
    return ch;
}

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

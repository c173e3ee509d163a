// vdl_fuse.h -- pattern fusion: recognises "select -> gather -> global fold" programs
// (the shape mplan2vdl emits for filter + aggregate queries such as TPC-H Q6:
// Select -> FoldSelect + Gather, /root/reference/src/Vlite.hs:721-730; ungrouped
// aggregate -> Fold over an all-zeros key, Vlite.hs:636-639,1048-1060) and turns them
// into one read-once scan per table: per-column closed-range filters AND-ed together,
// and per-aggregate products of affine column factors.
#pragma once
#include "vdl_ir.h"
#include "vdl_scan_desc.h"

#include <map>
#include <memory>
#include <string>
#include <vector>

namespace vdl {

struct ScanColumn {
    std::string name;          // catalog key path (VC_DIRECT / VC_GATHER / VC_INRANGE)
    int64_t lo = INT64_MIN;    // row passes iff lo <= value <= hi for every column
    int64_t hi = INT64_MAX;
    int kind = VC_DIRECT;
    int idx = -1, idx2 = -1;   // source virtual column(s): always earlier in the list
    int prelude = -1;
    std::vector<FormStep> form; // VC_FORM
    // the earlier columns this one is computed from
    std::vector<int> sources() const {
        std::vector<int> o;
        if (kind == VC_FORM) { for (const FormStep &f : form) if (f.op == FormStep::LEAF) o.push_back(f.col); return o; }
        if (idx >= 0) o.push_back(idx);
        if (idx2 >= 0) o.push_back(idx2);
        return o;
    }
};
// Work on the dimension side that has to happen before a scan with derived columns: the bitmap of a dimension-side
// selection (the validity of a `witness` statement, run by the per-operator executor: dimension filters, joins of
// dimensions with further dimensions), or the 0 / 1 table of a LIKE pattern over every offset of a string heap.
struct PreludeItem {
    enum Kind : int { DIM_BITMAP = 0, LIKE_LUT = 1, SEMI_BITMAP = 2 } kind = DIM_BITMAP;
    int witness = 0;           // DIM_BITMAP: statement whose vector is EPS exactly where the dimension selection rejects the row
    std::string heap, pattern; // LIKE_LUT
    // DIM_BITMAP whose selection is itself a conjunction of range filters over columns of the dimension table and lookups
    // through ITS foreign keys (orders filtered by date and by the customer's segment): one scan over the dimension table
    // writes the bitmap -- the select pass of the fused front with nothing to take -- instead of the per-operator executor
    // running the witness statement and everything under it.
    bool scan = false;
    std::string table;
    std::vector<ScanColumn> cols;
    bool never = false;
    // SEMI_BITMAP: the set {index(r) : r a selected row of `table`} as a bitmap over the rows of another table -- the semi-join the
    // compiler writes as Scatter(ones, fk mod N) + FoldSelect (LeftSemi with the dimension on the left, /root/reference/src/Vlite.hs:
    // 1212-1222; TPC-H Q4's EXISTS).  One scan of `table` (`cols`: its filters and the index atom, column `index_col`) sets the
    // bits with atomic ORs; the scan of the other table tests the bit of its own row id (VC_ROWID -> VC_BITS).  `modulus` is the N of
    // the emitted `mod`: exact only while the other table has at most N rows (checked when the plan is bound; else statement by statement)
    int index_col = -1;
    int64_t modulus = 0;
    std::string bits_of;       // a column of the table whose rows the bits stand for (its length)
};

struct ScanFactor {            // (a + s * column[col])
    int col = -1;
    int64_t a = 0, s = 1;
};

struct ScanAgg {
    int kind = AGG_SUM;
    int64_t constant = 1;      // used when there are no factors: the datum is this constant
    std::vector<ScanFactor> fac;
};

// Scalar expression over the aggregates of one scan, evaluated at finalisation
// (e.g. avg = Divide(FoldSum, FoldSum), /root/reference/src/Vlite.hs:1038-1041).
struct Scalar {
    enum K { AGG, CONST, BIN } k = CONST;
    int agg = -1;
    int64_t c = 0;
    int bin = -1;
    std::shared_ptr<const Scalar> l, r;
};
using ScalarP = std::shared_ptr<const Scalar>;

struct ScanPlan {
    std::string table;
    std::vector<ScanColumn> cols;
    std::vector<ScanAgg> aggs;   // partial words of this scan: [selected-row count, agg 0, agg 1, ...]
    bool never = false;          // predicate is constant false
};

// KeyStep program -> components, when it has that shape: returns their number (0 = not of the shape; the mask, if
// any, in *masked / *mask).  VDL_NO_CANON_KEY in the environment makes it answer 0.
int composite_key(const KeyStep *steps, int n, KeyComp *comps, int *masked, int64_t *mask);

struct GroupScanPlan {
    std::string table;
    std::vector<ScanColumn> cols;
    std::vector<KeyStep> key;
    int64_t pmin = 0, pcount = 0;    // bucket = key - pmin, must lie in [0, pcount)
    std::vector<ScanAgg> aggs;       // AGG_FIRST: fac[0].col = the column whose first-row value is taken
    bool never = false;
};

struct FusedOutput {
    int node = 0;                // MaterializeCompact id
    int scan = -1;               // index into FusedPlan::scans (global fold), or
    int gscan = -1;              // index into FusedPlan::gscans (one value per non-empty group)
    ScalarP value;
};

// A first-level filter (FoldSelect of a predicate over unfiltered table columns) whose predicate is a conjunction of
// per-column interval sets: it can be evaluated straight off the columns in one pass, whatever the rest of the
// program looks like (plans with joins do not fuse as a whole, but their Select steps do).
constexpr int kMaxFilterCols = 6, kMaxFilterIvs = 4;
struct FilterColumn { std::string name; int n = 0; int64_t lo[kMaxFilterIvs] = {}, hi[kMaxFilterIvs] = {}; };
struct FilterSpec { std::string table; std::vector<FilterColumn> cols; bool never = false; };

// A plan that does not fuse as a whole (its Partition has a sparse domain: joins feeding a GROUP BY, TPC-H Q3) can still
// have the front of its fact side fused: ONE scan evaluates the Select steps and the FK-join filters of the fact table and
// writes, for the surviving rows only, the columns the rest of the program reads -- fact columns, dimension columns looked
// up through the join index, row ids -- as sparse vectors on one shared selection.  The per-operator executor then starts
// at those statements (they are handed to it ready-made) instead of running the ~15 filter / gather statements that
// produce them one kernel at a time over the whole fact table.
struct ProjPlan {
    bool ok = false;
    std::string table, why;
    std::vector<ScanColumn> cols;    // as in ScanPlan (filters, derived columns)
    bool never = false;
    std::vector<int> nodes;          // statements the scan produces ...
    std::vector<int> node_col;       // ... and the column each one is (-1: the row ids themselves; -2 - e: the row expression exprs[e])
    std::vector<std::vector<KeyStep>> exprs;     // element-wise expressions over the columns that the scan evaluates for the survivors
};

struct FusedPlan {
    ProjPlan proj;
    std::vector<PreludeItem> prelude;    // dimension-side work the scans' derived columns need (empty for single-table plans)
    std::map<int, FilterSpec> filters;   // FoldSelect statement id -> its column filter (filled whether or not the plan fuses)
    bool ok = false;
    std::string why_not;         // reason the program did not fuse (reported by describe)
    std::vector<ScanPlan> scans;
    std::vector<GroupScanPlan> gscans;
    std::vector<FusedOutput> outputs;
};

void rewrite_program(Program &P);            // equivalent forms the planner and the sharded routes prefer (vdl_fuse.cpp)
FusedPlan fuse_program(const Program &P);
std::string describe_fused(const FusedPlan &F);
int64_t eval_scalar(const Scalar &s, const int64_t *agg_values);

}  // namespace vdl

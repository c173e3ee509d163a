// vdl_kernels.h -- launch wrappers for the HIP kernels (gfx950): vdl_kernels.hip (fused scan, column generator),
// vdl_mscan.hip (multi-aggregate / grouped scan), vdl_ops.hip (per-operator kernels), vdl_partition.hip.
// All launchers are asynchronous on the given stream and return the hipError_t of the launch.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "vdl_fuse.h"

namespace vdl {

// ---- operand descriptor of the per-operator kernels --------------------------------
// A vector operand is a column of 1/2/4/8-byte signed integers, or a virtual range
// (RangeV / RangeC never touch memory: /root/reference/src/Vdl.hs:428-434 notes the
// backend "only materializes what's needed").
enum SrcKind : int { SRC_I64 = 0, SRC_I32 = 1, SRC_I16 = 2, SRC_I8 = 3, SRC_RANGE = 4 };
struct Src {
    const void *p = nullptr;
    int kind = SRC_I64;
    int64_t from = 0, step = 0;
};

// ---- fused scan ----------------------------------------------------------------------
struct ScanArgs {
    int ncol = 0, nagg = 0;
    int64_t n = 0;
    const void *ptr[kMaxScanCols] = {};
    int width[kMaxScanCols] = {};            // bytes: 1, 2, 4, 8
    int filtered[kMaxScanCols] = {};         // 0: column has no range filter
    int64_t lo[kMaxScanCols] = {}, hi[kMaxScanCols] = {};
    int kind[kMaxScanAggs] = {};             // AGG_SUM / AGG_MIN / AGG_MAX
    uint32_t used[kMaxScanAggs] = {};        // bit c: column c contributes a factor
    uint32_t plain[kMaxScanAggs] = {};       // bit c: that factor is the bare column (a=0, s=1)
    int64_t fa[kMaxScanAggs][kMaxScanCols] = {}, fs[kMaxScanAggs][kMaxScanCols] = {};
    int64_t constant[kMaxScanAggs] = {};     // datum when the aggregate has no column factor
    int64_t *block_partials = nullptr;       // [grid][1 + nagg]
    int never = 0;
    int chunked = 0;                         // tile->block mapping: 0 grid-stride, 1 contiguous chunks
};

struct ScanLaunch { int grid = 0, block = 0, variant = 0; };
// Chooses template instantiation and grid for (ncol, nagg) on this device.
ScanLaunch scan_launch_config(ScanArgs &a, int num_cus);
hipError_t launch_scan(const ScanArgs &a, const ScanLaunch &cfg, hipStream_t s);
// Reduces the per-block partials into words[0..nagg] = {count, agg0, ...}.
hipError_t launch_scan_finish(const int64_t *block_partials, int nblocks, int nagg, const int *kinds_dev_or_null,
                              const ScanArgs &a, int64_t *words, hipStream_t s);
const char *scan_kernel_name(const ScanLaunch &cfg);

// ---- multi-aggregate fused scans (vdl_mscan.hip): global and grouped (dense-domain GROUP BY) ----
ScanLaunch mscan_launch_config(const MScanCols &cols, MScanDesc &d, bool grouped, int num_cus);
const char *mscan_kernel_name(const ScanLaunch &cfg);
// out: global form 1 + nagg words {row count, aggregates}; grouped form pcount * (nagg + 1) + 1 words: per
// bucket {row count, aggregates...}, last word = rows whose key fell outside [pmin, pmin + pcount).
// resolve_first: turn AGG_FIRST row ids into column values (single rank only).
// jit_fn: the scan kernel specialised for this plan (vdl_jit.cpp) instead of the precompiled variant cfg names
hipError_t launch_mscan(const MScanCols &cols, const MScanDesc &d, const MScanDesc *dev_desc, const ScanLaunch &cfg, bool grouped,
                        bool never, int64_t *out, bool resolve_first, hipStream_t s, hipFunction_t jit_fn = nullptr);
// for the specialiser: the by-value arguments, the chosen variant's shape, the dynamic LDS of a launch
MsArgs mscan_args(const MScanCols &cols);
void mscan_variant_shape(const ScanLaunch &cfg, int *nc, int *u, bool *vec, bool *grouped, bool *der);
size_t mscan_lds_bytes(const MScanDesc &d, bool grouped);
// Projection scans (ProjPlan, vdl_fuse.h).  launch_project_select: a selection over a table's rows as a bitmap (d.bitmap_only = 1:
// a dimension scan, out_ptr[0] = the bitmap) or as bits set in a semi-join set (d.bitmap_only = 2).
int64_t project_tiles(int64_t n);
hipError_t launch_project_select(const MScanCols &cols, const MScanDesc *dev_desc, int num_cus, hipStream_t s, hipFunction_t jit_fn = nullptr);
bool project_select_vec(const MScanCols &cols);          // the 16-byte-load form applies (alignment of the deciding columns)
// the fused front in ONE pass (vdl_mscan_body.h: project_front_body): scols / dev_sdesc = the deciding columns as the select pass saw
// them (out_ptr[0] = the selection's bitmap or null), tcols / dev_tdesc = every column with the output vectors (out_idx, out_ptr, out_cap);
// look = project_look_bytes(n) bytes of scratch; the survivors' number is left at total_dev and (if not null) in pinned total_host
int64_t project_look_bytes(int64_t n);
hipError_t launch_project_front(const MScanCols &scols, const MScanDesc *dev_sdesc, const MScanCols &tcols, const MScanDesc *dev_tdesc, void *look,
                                int64_t *total_dev, int64_t *total_host, int num_cus, hipStream_t s, hipFunction_t jit_fn = nullptr);
// sharded FoldChoose: after the MIN all-reduce of the row-id words, the owning rank substitutes the value, others 0
hipError_t launch_mscan_resolve_first(const MScanCols &cols, const MScanDesc &d, const MScanDesc *dev_desc, int64_t *table, hipStream_t s);

// ---- synthetic data --------------------------------------------------------------------
hipError_t launch_gen_column(void *out, int elem_bytes, int64_t row0, int64_t n, uint64_t seed, uint64_t col_id,
                             int64_t lo, int64_t hi, int64_t mul, int64_t add, hipStream_t s);

// ---- per-operator kernels -----------------------------------------------------------------
// validity bitmaps: bit (i & 63) of word (i >> 6); nullptr = every slot holds a value.
hipError_t launch_binary(int op, Src a, Src b, int64_t *out, int64_t n, hipStream_t s);
// A tree of element-wise operators over up to kExprLeaves vectors of one length, in postfix order, evaluated in one
// pass (the general path fuses chains of single-use Binary statements into this instead of one kernel per operator).
constexpr int kExprLeaves = 12, kExprInstrs = 48, kExprDepth = 8;
// operators that exist only inside fused trees: the emitter's sugar (a >= b is printed as LogicalOr(Greater(a,b), Equals(b,a)),
// /root/reference/src/Vdl.hs:139-152) recognised back, one interpreted instruction instead of three
enum : int { X_GE = 32, X_NE = 33 };
struct ExprProg {
    int n_instr = 0, n_leaf = 0;
    Src leaf[kExprLeaves];
    signed char code[kExprInstrs] = {};      // >= 0: apply BinOp code to the two topmost values; < 0: push leaf (-code - 1)
};
hipError_t launch_expr(const ExprProg &prog, int64_t *out, int64_t n, hipStream_t s);
hipError_t launch_poison(void *p, size_t bytes, uint64_t seed, hipStream_t s);     // VDL_POISON=rand (debugging)
hipError_t launch_and_words(const uint64_t *a, const uint64_t *b, uint64_t *out, int64_t nwords, hipStream_t s);
// A filter predicate -- comparisons between stored vectors joined by LogicalAnd / LogicalOr -- evaluated straight into
// the selection bitmap: a comparison of 64 rows is one 64-bit mask (v_cmp writes it), the connectives work on masks.
// Postfix: code >= 0 pushes comparison #code, P_AND / P_OR combine the two topmost masks.
constexpr int kPredCmps = 16, kPredInstrs = 32, kPredDepth = 8;
enum : int { P_GT = 0, P_EQ = 1, P_GE = 2, P_NE = 3, P_AND = -1, P_OR = -2 };
struct PredProg {
    int n_instr = 0, n_cmp = 0;
    Src a[kPredCmps], b[kPredCmps];
    signed char op[kPredCmps] = {};
    signed char code[kPredInstrs] = {};
};
// out bit i = predicate(i) & valid bit i (valid null = every slot)
hipError_t launch_pred(const PredProg &prog, const uint64_t *valid, uint64_t *out, int64_t n, hipStream_t s);
// global fold record {value, first slot, count} <-> three words that merge across shards (reduce: 0 sum, 1 min, 2 max)
// Merge of the partial words gathered from all ranks (vdl_comm.cpp).  Layout: rank r's block is 2 * n_words int64 at
// g + r * 2 * n_words -- the words, then the words again with FoldChoose (VDL_REDUCE_FIRST) entries resolved to the value
// at that rank's own smallest row id.  SUM / MIN / MAX fold the first halves; FIRST takes the value of the rank holding
// the smallest global row id (0 when no rank has a row in the group).  Host and device use this one definition.
#if defined(__HIPCC__)
#define VDL_HD __host__ __device__
#else
#define VDL_HD
#endif
// (`stride`: int64 words per rank in `g`: 2 * n_words, plus whatever the caller keeps behind them -- the status word of the
// sharded fold route)
VDL_HD inline int64_t merge_word(const int64_t *g, int world, int64_t n_words, int op, int64_t i, int64_t stride) {
    if (op == 4 /* VDL_REDUCE_FIRST */) {
        int64_t best = INT64_MAX, val = 0;
        for (int r = 0; r < world; r++) {
            const int64_t id = g[(int64_t)r * stride + i];
            if (id < best) { best = id; val = g[(int64_t)r * stride + n_words + i]; }
        }
        return val;
    }
    int64_t acc = g[i];
    for (int r = 1; r < world; r++) {
        const int64_t x = g[(int64_t)r * stride + i];
        if (op == 2 /* MIN */) acc = x < acc ? x : acc;
        else if (op == 3 /* MAX */) acc = x > acc ? x : acc;
        else acc = (int64_t)((uint64_t)acc + (uint64_t)x);
    }
    return acc;
}
// gathered: `stride` words per rank = n_words raw, n_words resolved, then (stride > 2 * n_words) the rank's status word.
// status_out (may be null; needs that word): [0] = the first non-zero status over the ranks (0 = every rank's local phase
// succeeded), [1] = that rank.
// OR of the ranks' semi-join sets: gathered holds, per rank, `words` set words then ONE word = the rows of the set's source table on
// that rank; out = the union, with the bits at and beyond the table's GLOBAL length (the sum of those words) cleared -- the Scatter
// the set stands for is that long, positions beyond it are dropped (Vlite.hs:1212-1222).
hipError_t launch_or_sets(const uint64_t *gathered, int world, int64_t words, uint64_t *out, hipStream_t s);
hipError_t launch_merge_words(const int64_t *gathered, int world, int64_t n_words, int64_t stride, const int32_t *ops, int64_t *out,
                              int64_t *status_out, hipStream_t s);
hipError_t launch_fold_words(const int64_t *rec, int reduce, int64_t row0, int64_t *out, hipStream_t s);
hipError_t launch_fold_record(const int64_t *words, int64_t *rec, hipStream_t s);
// Gather out of a sparse vector without densifying it: counts[w] = popcount of bitmap word w (the caller turns them into
// exclusive prefix sums = wrank), then out[i] = entries[wrank[p/64] + popcount(word below bit p)] for p = pos[i] selected.
hipError_t launch_word_counts(const uint64_t *bitmap, int64_t nwords, int64_t *counts, hipStream_t s);
hipError_t launch_gather_ranked(const int64_t *entries, const uint64_t *bitmap, const int64_t *wrank, int64_t nsrc, Src pos, const uint64_t *vpos,
                                int64_t n, int64_t *out, uint64_t *vout, hipStream_t s);
// A first-level filter evaluated straight off its columns: bit i = every column's value lies in one of its intervals
struct FilterArgs {
    int ncol = 0, never = 0;
    Src col[kMaxFilterCols];
    int nint[kMaxFilterCols] = {};
    int64_t lo[kMaxFilterCols][kMaxFilterIvs] = {}, hi[kMaxFilterCols][kMaxFilterIvs] = {};
};
hipError_t launch_filter_columns(const FilterArgs &a, uint64_t *out, int64_t n, hipStream_t s);
// FoldSelect with unit-length runs: out bitmap = (d != 0) & vd & vc
hipError_t launch_select_bitmap(Src d, const uint64_t *vd, const uint64_t *vc, uint64_t *out, int64_t n, hipStream_t s);
// Global (single-run) fold of `d` over slots valid in both bitmaps.
// result[0] = value, result[1] = first slot whose control is valid (or -1), result[2] = number of data folded.
// scratch: at least 3 * fold_scratch_blocks() int64.
int fold_scratch_blocks();
hipError_t launch_fold_global(int kind /*0 sum,1 min,2 max,3 count,4 choose*/, Src d, const uint64_t *vd, const uint64_t *vc,
                              int64_t n, int64_t *scratch, int64_t *result, hipStream_t s);
// element-wise op on two one-hot (fold result) vectors: valid iff both valid and same slot
hipError_t launch_onehot_binary(int op, const int64_t *a, const int64_t *b, int64_t *out, hipStream_t s);
// broadcast-constant operand: b_is_const selects (a op k) / (k op a)
hipError_t launch_onehot_const(int op, const int64_t *a, int64_t k, int const_left, int64_t *out, hipStream_t s);
// one-hot -> dense n-slot vector (values + bitmap)
hipError_t launch_onehot_dense(const int64_t *oh, int64_t *out, uint64_t *valid, int64_t n, hipStream_t s);

// MaterializeCompact: out[rank(i)] = v[i] for valid i, slot order kept.
// block_counts: ceil(n / compact_tile()) int64; returns total in block_counts[nblocks] (device).
int64_t compact_tile();
hipError_t launch_compact_count(const uint64_t *valid, int64_t n, int64_t *block_counts, hipStream_t s);
hipError_t launch_compact_scan(int64_t *block_counts, int64_t nblocks, hipStream_t s, int64_t *host_total = nullptr);
// k words of device memory into pinned host memory and then `seq` into the pinned flag word, with system-scope stores (the host polls)
hipError_t launch_post_words(const int64_t *src, int64_t k, int64_t *pinned_dst, int64_t *pinned_flag, int64_t seq, hipStream_t s);   // exclusive scan in place, total at [nblocks]
hipError_t launch_compact_offsets(const uint64_t *valid, int64_t n, int64_t *block_counts, hipStream_t s, int64_t *host_total = nullptr);   // host_total: pinned host word that also receives the total   // count + scan (one launch for small n)
hipError_t launch_compact_write(Src v, const uint64_t *valid, int64_t n, const int64_t *block_offsets, int64_t *out, hipStream_t s, int64_t *out_pos = nullptr /* also: the slot number of every packed entry */);

// Gather / Scatter / Partition / segmented folds
hipError_t launch_gather(Src src, const uint64_t *vsrc, int64_t nsrc, Src pos, const uint64_t *vpos, int64_t n,
                         int64_t *out, uint64_t *vout, hipStream_t s);
hipError_t launch_scatter(Src src, const uint64_t *vsrc, Src pos, const uint64_t *vpos, int64_t n, int64_t nout,
                          int64_t *out, uint64_t *vout /* pre-zeroed; null = not wanted */, hipStream_t s);
hipError_t launch_fill_words(uint64_t *p, uint64_t v, int64_t nwords, hipStream_t s);
// FoldSelect (unit runs) straight over Gather(src, pos): out bit i = vc[i] & (src[pos[i]] exists and != 0); the gathered vector is never stored
hipError_t launch_select_gather(Src src, const uint64_t *vsrc, int64_t nsrc, Src pos, const uint64_t *vpos, const uint64_t *vc, uint64_t *out,
                                int64_t n, hipStream_t s);
// bitmap[idx[k]] = 1 for k < m (bitmap pre-zeroed; idx ascending, so neighbours often share a word)
hipError_t launch_set_bits(const int64_t *idx, int64_t m, uint64_t *bitmap, hipStream_t s, int64_t nbits = INT64_MAX /* entries outside [0, nbits) set nothing */);

// FoldSelect over general runs (vdl_engine.cpp: fold_select_runs): head flags of the runs of `ctl` (m entries, all valid),
// then the sort key 2 * run + (entry not selected) and the bitmap of selected entries
hipError_t launch_run_heads(const int64_t *ctl, int64_t m, int64_t *flags, int64_t *flags_copy, hipStream_t s);
hipError_t launch_fsel_keys(const int64_t *excl_heads, const int64_t *flags, const int64_t *d, const uint64_t *vd, int64_t m,
                            int64_t *keys, uint64_t *selected, hipStream_t s);

// device-wide exclusive prefix sum (in place); sums = prefix_sum_blocks(n) + 1 int64 of scratch
int64_t prefix_sum_blocks(int64_t n);
hipError_t launch_prefix_sum(int64_t *x, int64_t n, int64_t *sums, hipStream_t s);

// Partition with pivots RangeC pmin pcount 1 (bucket = clamp(data - pmin, 0, pcount)); see vdl_partition.hip
int64_t partition_tiles(int64_t n);
int partition_passes(int64_t pcount);
// flag[0] (pre-zeroed by the caller) becomes 1 when d[i] > d[i+1] somewhere: the data is not in non-decreasing order
hipError_t launch_sorted_check(Src d, int64_t n, int64_t *flag, hipStream_t s);
// The same pass also leaves the RUN HEADS of the (fully valid) vector: heads bit i = (i == 0 or d[i] != d[i-1]) -- what the folds of
// the GROUP BY that this Partition serves need when the data turns out to be in order (their control vector is then this
// vector, unmoved).  flag[0] / flag[1] as above (pre-set to 0 / INT64_MIN).
hipError_t launch_sorted_heads(Src d, int64_t n, uint64_t *heads /* (n + 63) / 64 words */, int64_t *flag /* 3 words: descends, largest, smallest */, hipStream_t s);
// ... with the heads counted per compaction tile, the counts scanned in place (counts[nb] = total, then {descends, largest, smallest}) and
// those four words posted into pinned host memory behind `seq` -- one launch.  `state`: words owned by the caller between launches,
// once prepared by launch_sorted_state_init.  Serves int64 vectors at 16-byte aligned addresses (sorted_heads_counted_serves).
bool sorted_heads_counted_serves(Src d, int64_t n);
int64_t sorted_heads_state_words();
int64_t sorted_heads_counts_words(int64_t n);
hipError_t launch_sorted_state_init(int64_t *state /* sorted_heads_state_words() */, hipStream_t s);
hipError_t launch_sorted_heads_counted(const int64_t *d, int64_t n, uint64_t *heads, int64_t *counts /* sorted_heads_counts_words(n) */, int64_t *state, int64_t *pinned_dst /* 4 words */,
                                       int64_t *pinned_flag, int64_t seq, hipStream_t s);
// Every fold of one GROUP BY in one launch, results PACKED (one per run, in run order): fold j reduces data[j] over the runs whose
// heads are given (m entries, all holding a value; entry 0 is a head) and writes out[j][g] for the g-th run.  offsets = exclusive
// prefix of the head counts per compaction tile (launch_compact_count + launch_compact_scan over `heads`).  kind: 0 sum, 1 min,
// 2 max, 3 count, 4 choose (first value).  out[j] of kinds 0-3 must hold the reduction's identity (0 / INT64_MAX / INT64_MIN / 0).
constexpr int kMaxGroupFolds = 16;
struct GroupFoldArgs {
    int nfold = 0;
    int kind[kMaxGroupFolds] = {};
    Src data[kMaxGroupFolds];
    int64_t *out[kMaxGroupFolds] = {};
};
hipError_t launch_group_fold(const GroupFoldArgs &a, const uint64_t *heads, int64_t m, const int64_t *offsets, hipStream_t s);
size_t partition_scratch_bytes(int64_t n, int64_t pcount);
hipError_t launch_partition(Src data, const uint64_t *valid, int64_t n, int64_t pmin, int64_t pcount, void *scratch /* partition_scratch_bytes(n, pcount) */,
                            uint64_t *keys_a, int64_t *slots_a, uint64_t *keys_b, int64_t *slots_b,
                            int64_t *n_valid_dev, int64_t *pos_out, hipStream_t s, int64_t max_bucket = -1 /* largest bucket known to occur */,
                            int64_t *order_out = nullptr /* instead of pos_out: the slots in rank order (order[pos[slot]] = slot), stored sequentially */,
                            int64_t *sorted_keys_out = nullptr /* with order_out: bucket + pmin in rank order (= the keys, when all lie inside the pivots) */);

// Fold over a general control vector; kind 0 sum, 1 min, 2 max, 3 count, 4 choose
// heads: nwords uint64; wordhd: nwords + maxscan_blocks(nwords) int64
int64_t maxscan_blocks(int64_t n);
hipError_t launch_fold_heads(Src ctl, const uint64_t *vc, int64_t n, uint64_t *heads, int64_t *wordhd, hipStream_t s);
hipError_t launch_fold_runs(int kind, Src d, const uint64_t *vd, const uint64_t *vc, const uint64_t *heads, const int64_t *wordhd, int64_t n,
                            int64_t *out, uint64_t *vout, hipStream_t s);
hipError_t launch_fold_segmented(int kind, Src ctl, const uint64_t *vc, Src d, const uint64_t *vd, int64_t n, uint64_t *heads,
                                 int64_t *wordhd, int64_t *out, uint64_t *vout, hipStream_t s);

// ---- row exchange for sharded Partition (see vdl_partition.hip) ------------------------------------
constexpr int kMaxExSources = 62;
constexpr int kMaxExWorld = 128;          // ranks in one exchange (vdl_exchange_begin)
// CrossProductOuter / Inner (Vdl.hs:412-416; Vlite.hs:278-289): positions i / k and i % k over n = m * k slots
hipError_t launch_cross(int64_t n, int64_t k, int inner, int64_t *out, hipStream_t s);
// Like (Vdl.hs:444-447): out[i] = string at byte offset data[i] of `heap` matches the SQL LIKE pattern
struct LikePattern { unsigned char p[256]; int len; };
hipError_t launch_like(Src data, const uint64_t *vdata, int64_t n, Src heap, const uint64_t *vheap, int64_t heap_n, const LikePattern &pat,
                       int64_t *out, hipStream_t s);
constexpr int kExBins = 4096;              // at most this many equal, power-of-two-wide slices of the pivots' domain in which a rank counts its keys (hist: kExBins + 1 words, the last = keys outside the pivots; pre-zeroed)
hipError_t launch_ex_hist(Src key, const uint64_t *vkey, int64_t n, int64_t pmin, int64_t pcount, int64_t *hist, hipStream_t s);
// Routing of a sharded Partition's rows (vdl_partition.hip "routing"): where every row goes, the rows per destination, and the packed send buffer.
struct ExRoute {
    Src key; const uint64_t *vkey = nullptr;
    int64_t n = 0, pmin = 0, pcount = 0;                    // pcount <= 0: every row with a key takes part and goes to destination 0
    int world = 1, shift = 0;                               // shift = ex_route_shift(pcount): the width of a slice of the domain
    const int32_t *owner = nullptr;                         // slice of the domain -> rank (kExBins entries; the balanced cut of vdl_run_sharded), or null: the domain cut evenly
};
constexpr int kExPackCols = 8;             // columns one launch of the pack writes
struct ExCols {
    int ncol = 0, first = 0;               // columns first .. first + ncol - 1 of the send buffer
    Src src[kExPackCols];
    int mask_at = -1, nvalid = 0;          // mask_at >= 0: this launch also writes the mask column there: bit v = valid[v] holds a value in the row
    const uint64_t *valid[kMaxExSources] = {};
};
int64_t ex_route_tiles(int64_t n);
int ex_route_shift(int64_t pcount);
hipError_t launch_ex_route(const ExRoute &R, int64_t *tileoff /* world x ex_route_tiles(n) */, int64_t *counts /* 2 world + 1, pre-zeroed: rows per destination, keys outside the pivots, where each destination's rows begin */, hipStream_t s);
hipError_t launch_ex_pack_all(const ExRoute &R, const ExCols &C, const int64_t *tileoff, const int64_t *counts /* as launch_ex_route left them */, int64_t n_send, int64_t *out, hipStream_t s);
hipError_t launch_ex_unmask(const int64_t *mask, int64_t n, int j, uint64_t *valid, hipStream_t s);

}  // namespace vdl

// vdl_scan_desc.h -- what the fused-scan kernels are told about a scan: limits, column kinds, key / condition steps, the
// descriptor (MScanDesc) and the by-value arguments (MsArgs).  Free of the C++ library: this text is also handed to hiprtc
// when a plan's kernels are specialised at run time (vdl_jit.cpp), where only the HIP built-ins exist.
#pragma once
#if defined(__HIPCC_RTC__)
typedef signed char int8_t;
typedef short int16_t;
typedef int int32_t;
typedef long long int64_t;
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef unsigned long long uint64_t;
typedef unsigned long long uintptr_t;
#define INT64_MAX 0x7fffffffffffffffLL
#define INT64_MIN (-INT64_MAX - 1)
#else
#include <cstdint>
#endif

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define VDL_SD __host__ __device__
#else
#define VDL_SD
#endif

namespace vdl {

// element-wise binary operators, /root/reference/src/Vdl.hs:110-122
enum BinOp : int {
    B_LAND, B_LOR, B_BAND, B_BOR, B_SHIFT, B_EQ, B_ADD, B_SUB, B_GT, B_MUL, B_DIV, B_MOD, B_COUNT
};

constexpr int kMaxScanCols = 8;          // plain scans (table columns only)
constexpr int kMaxJoinScanCols = 12;     // scans with derived columns (lookups, differences, formulas): each one is a column too
constexpr int kMaxScanAggs = 8;
constexpr int kMaxFactors = 4;

enum AggKind : int { AGG_SUM = 0, AGG_MIN = 1, AGG_MAX = 2, AGG_FIRST = 3 };   // FIRST: value of a column at the group's first row

// A scan reads table columns row by row (VC_DIRECT) and may derive further per-row values from them -- the FK-join
// lowering of the compiler (/root/reference/src/Vlite.hs:1199-1282) seen from the fact table: the dimension side of a
// join is a lookup through the join-index column, so a fact-table scan whose extra "columns" are dim_col[fk[row]] or
// dim_bitmap[fk[row]] evaluates the join, its filters and its aggregates in one pass.
enum VColKind : int {
    VC_DIRECT = 0,   // name = catalog column of the scanned table
    VC_GATHER = 1,   // value = column `name` (of another table) at row v[idx]; v[idx] outside that column -> the row is EPS
    VC_BITS = 2,     // value = bit v[idx] of a dimension-side selection bitmap (FusedPlan::prelude[prelude]): 0 / 1; outside -> EPS
    VC_LUT = 3,      // value = prelude[prelude] (a lookup table) at v[idx]; outside the table -> 0 (Like over heap offsets)
    VC_INRANGE = 4,  // value = 1; the row is EPS unless 0 <= v[idx] < rows of column `name` (a Gather out of an unfiltered table)
    VC_SUB = 5,      // value = v[idx] - v[idx2] (column against column comparisons become a range filter on the difference)
    VC_ROWID = 7,    // value = the row's own index in the scanned table (the index of a lookup into a set built over this table's rows:
                     // the semi-join bitmap of PreludeItem::SEMI_BITMAP)
    VC_FORM = 6      // value = 0 / 1: a boolean formula over range tests of earlier columns (ScanColumn::form) -- IN lists, disjunctions
                     // across columns (Q19), CASE WHEN conditions used as aggregate inputs (Q12, Q14)
};
// One step of a formula in postfix order, evaluated per row on a stack of bits: LEAF pushes lo <= v[col] <= hi.
struct FormStep {
    enum Op : int { LEAF = 0, AND = 1, OR = 2, NOT = 3, TRUE_ = 4, FALSE_ = 5,
                    REF = 6 };   // kernel layout only (MScanDesc::form): push the result of test number `col`
    int op = LEAF, col = -1;
    int64_t lo = 0, hi = 0;
};
constexpr int kMaxFormSteps = 64, kMaxFormDepth = 30;   // (at most 64 tests per formula: their results are the bits of one word)
constexpr int kMaxFormPool = 160;                       // all formula columns of one scan, tests + programs
// ---- grouped scan (dense-domain GROUP BY): the key program ------------------------------------
// The group key is evaluated per row by a two-accumulator straight-line program:
//   acc / tmp <- column, then (op constant) steps on either, and `acc = acc op tmp` combines.
// This is exactly the shape makeCompositeKey emits (/root/reference/src/Vlite.hs:1123-1170):
// ((c0 >> tz0) - min0) << bits | ((c1 >> tz1) - min1) ... & mask.
constexpr int kMaxKeySteps = 32;
struct KeyStep {
    enum Kind : int { LOAD = 0, OPK = 1, COMBINE = 2 } kind = LOAD;
    int target = 0;          // 0 = acc, 1 = tmp (LOAD / OPK)
    int col = -1;            // LOAD: scan column index
    int bin = -1;            // OPK / COMBINE: BinOp
    int const_left = 0;      // OPK: result = k op x instead of x op k; COMBINE: acc = tmp op acc
    int64_t k = 0;
};

// A group key of the shape makeCompositeKey emits (Vlite.hs:1123-1170): OR over components ((col >> rsh) - sub) << lsh,
// optionally ANDed with a mask.  The grouped scan evaluates this form in straight-line code; any other key program is
// interpreted step by step (KeyStep).
constexpr int kMaxKeyComps = 4;
struct KeyComp { int col = 0, rsh = 0, lsh = 0, pad = 0; int64_t sub = 0; };
// (the projection scan's take pass handles up to 16 columns; its select pass sees only the columns that decide a row's survival,
// at most kMaxSelectCols of them)
constexpr int kMaxProjCols = 16, kMaxSelectCols = 12, kMaxProjOuts = 10;

constexpr int kMaxGroupAggs = 16;
constexpr int kMaxVCols = kMaxProjCols;      // columns of a scan descriptor: 8 for plain aggregate scans, up to 12 with derived columns, 16 for the projection scan
static_assert(kMaxVCols >= kMaxScanCols && kMaxVCols >= kMaxJoinScanCols, "the aggregate scans' columns fit the descriptor");
struct MScanCols {                           // host-side description of a scan's columns
    int ncol = 0;
    int64_t n = 0, row0 = 0;
    bool rowid_global = false;               // row-id columns count from the TABLE's first row (the sharded front route) instead of the shard's
    const void *ptr[kMaxVCols] = {};         // VC_DIRECT: the column; derived columns: the table looked up (column / bitmap words / LUT)
    int width[kMaxVCols] = {};
    int filtered[kMaxVCols] = {};
    int64_t lo[kMaxVCols] = {}, hi[kMaxVCols] = {};
    int kind[kMaxVCols] = {};                // VColKind (vdl_fuse.h); 0 = read from the scanned table
    int lazy[kMaxVCols] = {};                // projection scan: the column decides nothing about a row's survival -- read it for survivors only
};
struct MAggDesc {
    int kind = 0;                            // AGG_SUM / AGG_MIN / AGG_MAX / AGG_FIRST
    uint32_t used = 0, plain = 0;            // bit c: column c contributes a factor / the factor is the bare column
    int pad = 0;
    int64_t constant = 0;                    // datum when there is no column factor
    int64_t fa[kMaxProjCols] = {}, fs[kMaxProjCols] = {};
};
struct MScanDesc {                           // lives in device memory, read with scalar loads
    int nagg = 0, nkey = 0, replicas = 1, pad = 0;
    int64_t pmin = 0, pcount = 0;            // grouped: bucket = key - pmin in [0, pcount)
    int64_t *block_partials = nullptr;       // global: [grid][1 + nagg]; grouped: [grid][pcount * (1 + nagg) + 1]
    int64_t flo[kMaxVCols] = {}, fhi[kMaxVCols] = {};            // range filter per column (read only for filtered columns)
    int dkind[kMaxVCols] = {}, dsrc[kMaxVCols] = {}, dsrc2[kMaxVCols] = {};   // derived columns: VColKind, source column(s); VC_FORM: first step, steps
    // formula columns (VC_FORM), one after the other: column c owns form[dsrc[c] .. dsrc[c] + dsrc2[c]) -- first its dtests[c]
    // range tests sorted by column, then the postfix program over their results (FormStep::REF)
    int dtests[kMaxVCols] = {};
    FormStep form[kMaxFormPool];
    int64_t dn[kMaxVCols] = {};              // derived columns: entries of the table looked up
    // projection scan (k_project): what to write for the surviving rows
    int nout = 0, out_col[kMaxProjOuts] = {};     // out_col[o] >= 0: that column; -2 - e: the row expression e (below)
    // row expressions the take pass computes for the survivors instead of handing their operands over (Q3's composite group key
    // and its revenue term: the executor then starts at the Partition): expression e = steps key[expr_at[e] .. + expr_len[e]) of the
    // two-accumulator program form of the group keys (KeyStep), its value the accumulator at the end
    int nexpr = 0, expr_at[kMaxProjOuts] = {}, expr_len[kMaxProjOuts] = {};
    int64_t *out_ptr[kMaxProjOuts] = {};     // one packed int64 vector per produced column
    int64_t *out_idx = nullptr;              // the surviving rows' slot ids, ascending
    uint32_t take = 0;                       // k_project_take: the columns the outputs need (with the columns they are derived from)
    int bitmap_only = 0;                     // k_project_select: 1 = a dimension scan -- the selection's bitmap, no counts, no positions;
                                             // 2 = a semi-join scan: for every selected row, bit v[out_col[0]] of the set out_ptr[0] (dn[out_col[0]] bits) is set
    int ncomp = 0, key_masked = 0;           // ncomp > 0: the key program is this canonical form
    int64_t key_mask = 0;
    KeyComp comp[kMaxKeyComps];
    MAggDesc agg[kMaxGroupAggs];
    KeyStep key[kMaxKeySteps];
    // census builds of a staged scan (vdl_jit.cpp, VDL_CENSUS; measurement only, never the timed kernel): [column] = number of
    // distinct 128-byte lines the late loads of that column asked for
    unsigned long long *census = nullptr;
    // the fused front: columns its select side has in registers anyway (they decide survival) AND the outputs want for the survivors --
    // Q3's join index -- stay in LDS in survivor order (kFrontCarry entries per batch of tiles) instead of being fetched again line by
    // isolated line.  Bit c: column c (of this descriptor's own numbering) is carried; the i-th set bit of the select side's word and
    // of the take side's name the same column.  Survivors beyond the area's capacity fetch the column again.
    uint32_t carry = 0;
    // the fused front: entries the output vectors hold (the pass runs before the host knows the survivors' number, with room for a
    // guess: what does not fit is not written, and the host runs the pass again when it learns that the guess was short)
    int64_t out_cap = INT64_MAX;
};
constexpr int kMaxCarry = 2;

// What the scan kernels take by value: column bases (kept in the global address space), the widths and the
// filtered-column set packed into one word each.  (MScanCols itself held 64 SGPRs live across the tile loop -- widths,
// flags and sixteen 64-bit bounds -- and the grouped kernel spilled hundreds of scalar values into VGPR lanes; the
// bounds now sit in the device descriptor and are read where a column is actually filtered.)
struct MsArgs {
    int ncol = 0;
    uint64_t widths = 0;                     // 4 bits per column: bytes
    uint32_t filtered = 0;                   // bit c: column c has a range filter (MScanDesc::flo / fhi)
    uint32_t derived = 0;                    // bit c: column c is derived from earlier columns (MScanDesc::dkind ...), ptr[c] = its table
    uint32_t lazy = 0;                       // bit c: not read with the tile (projection scan: needed for surviving rows only; staged scans: below)
    // specialised aggregate scans that read late (vdl_jit.cpp): 4 bits per table column -- 0: read with the tile; 1..3: a filter
    // column read for the rows that passed the earlier stages; 14: a source of derived columns / of the group key, read for
    // the rows that passed every filter on table columns; 15: an aggregate input only, read for the rows that passed everything
    uint64_t stages = 0;
    VDL_SD constexpr int stage(int c) const { return (int)((stages >> (4 * c)) & 15u); }
    // the QUEUE form of a specialised scan that reads late (round 4): the columns outside `lazy` (the most selective filter column) come with the
    // tile and are filtered there; the rows still in are queued per wave and everything else is read for 64 of them at a time, every lane busy
    int queued = 0;
    int64_t n = 0, row0 = 0;
    int64_t rowid_base = 0;                  // what a row-id column subtracts from the global row number: row0 (ids inside this shard) or 0
    const void *ptr[kMaxVCols] = {};
    VDL_SD constexpr int width(int c) const { return (int)((widths >> (4 * c)) & 15u); }
};

}  // namespace vdl

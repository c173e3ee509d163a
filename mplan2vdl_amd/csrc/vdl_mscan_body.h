// vdl_mscan_body.h -- device code of the multi-aggregate fused scans (global and dense-domain grouped form, with or without
// derived columns): included by vdl_mscan.hip for the precompiled instantiations, and handed as text to hiprtc when a plan's
// scan is specialised at run time (vdl_jit.cpp).  HIP built-ins only -- no C++ library.
#pragma once
#include "vdl_scan_desc.h"

#ifndef VDL_SPEC_UNROLL
#define VDL_SPEC_UNROLL                     // specialised builds: _Pragma("unroll") on the loops a descriptor drives
#endif

namespace vdl {

typedef long long ll2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef char i8x2 __attribute__((ext_vector_type(2)));

template <int N> struct IntTag { static constexpr int value = N; };

namespace {


constexpr int kWave = 64;
constexpr int kMsBlock = 256;
enum { R_SUM = 0, R_MIN = 1, R_MAX = 2 };

__device__ __forceinline__ int rk_of(int kind) { return kind == AGG_SUM ? R_SUM : kind == AGG_MAX ? R_MAX : R_MIN; }
__device__ __forceinline__ int64_t r_identity(int rk) { return rk == R_SUM ? 0 : rk == R_MIN ? INT64_MAX : INT64_MIN; }
__device__ __forceinline__ int64_t r_combine(int rk, int64_t a, int64_t b) {
    if (rk == R_SUM) return (int64_t)((uint64_t)a + (uint64_t)b);
    if (rk == R_MIN) return a < b ? a : b;
    return a > b ? a : b;
}
__device__ __forceinline__ int64_t wave_reduce(int64_t x, int rk) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) x = r_combine(rk, x, __shfl_down(x, off, kWave));
    return x;
}
__device__ __forceinline__ int64_t load_scalar(const void *p, int width, int64_t i) {
    switch (width) {
    case 8: return ((const int64_t *)p)[i];
    case 4: return ((const int32_t *)p)[i];
    case 2: return ((const int16_t *)p)[i];
    default: return ((const int8_t *)p)[i];
    }
}
template <bool NT, typename V>
__device__ __forceinline__ V stream_load(const char *p) {
    if (NT) return __builtin_nontemporal_load((const V *)p);
    return *(const V *)p;
}

// key-program operators: apply_bin minus Divide / Modulo (the planner keeps those off this path)
__device__ __forceinline__ int64_t key_bin(int op, int64_t a, int64_t b) {
    switch (op) {
    case B_LAND: return (a != 0) && (b != 0);
    case B_LOR:  return (a != 0) || (b != 0);
    case B_BAND: return a & b;
    case B_BOR:  return a | b;
    case B_SHIFT:
        if (b >= 0) return a >> (b > 63 ? 63 : b);
        if (b <= -64) return 0;
        return (int64_t)((uint64_t)a << (unsigned)(-b));
    case B_EQ:  return a == b;
    case B_ADD: return (int64_t)((uint64_t)a + (uint64_t)b);
    case B_SUB: return (int64_t)((uint64_t)a - (uint64_t)b);
    case B_GT:  return a > b;
    default:    return (int64_t)((uint64_t)a * (uint64_t)b);
    }
}

// x[r] = x[r] op k (or k op x[r]) for all rows; the operator switch is wave-uniform and sits OUTSIDE
// the row loop (a per-row switch made the kernel scalar-issue bound)
#define VDL_ROWS(EXPR) { _Pragma("unroll") for (int r = 0; r < RW; r++) { const int64_t a = x[r]; x[r] = (EXPR); } }
template <int RW>
__device__ __forceinline__ void key_rows(int op, int const_left, int64_t (&x)[RW], int64_t k) {
    switch (op) {
    case B_BAND: VDL_ROWS(a & k) break;
    case B_BOR:  VDL_ROWS(a | k) break;
    case B_ADD:  VDL_ROWS((int64_t)((uint64_t)a + (uint64_t)k)) break;
    case B_MUL:  VDL_ROWS((int64_t)((uint64_t)a * (uint64_t)k)) break;
    case B_SUB:  if (const_left) VDL_ROWS((int64_t)((uint64_t)k - (uint64_t)a)) else VDL_ROWS((int64_t)((uint64_t)a - (uint64_t)k)) break;
    case B_SHIFT:
        if (const_left) VDL_ROWS(key_bin(B_SHIFT, k, a))
        else if (k >= 0) { const int sh = k > 63 ? 63 : (int)k; VDL_ROWS(a >> sh) }
        else if (k <= -64) VDL_ROWS(0 * a)
        else { const int sh = (int)(-k); VDL_ROWS((int64_t)((uint64_t)a << sh)) }
        break;
    default:
        if (const_left) VDL_ROWS(key_bin(op, k, a)) else VDL_ROWS(key_bin(op, a, k))
        break;
    }
}
#undef VDL_ROWS
template <int RW>
__device__ __forceinline__ void key_combine(int op, int swap, int64_t (&acc)[RW], const int64_t (&tmp)[RW]) {
    switch (op) {
    case B_BOR:
#pragma unroll
        for (int r = 0; r < RW; r++) acc[r] |= tmp[r];
        break;
    case B_BAND:
#pragma unroll
        for (int r = 0; r < RW; r++) acc[r] &= tmp[r];
        break;
    case B_ADD:
#pragma unroll
        for (int r = 0; r < RW; r++) acc[r] = (int64_t)((uint64_t)acc[r] + (uint64_t)tmp[r]);
        break;
    default:
#pragma unroll
        for (int r = 0; r < RW; r++) acc[r] = swap ? key_bin(op, tmp[r], acc[r]) : key_bin(op, acc[r], tmp[r]);
        break;
    }
}


template <int NC, int U, bool VEC, bool NT>
__device__ __forceinline__ void load_tile(const MsArgs &C, const MsArgs &Cr, int64_t base, int64_t (&v)[NC][2 * U], uint32_t skip = 0) {
    constexpr int BS = kMsBlock;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if (c < C.ncol && !(((C.derived | skip) >> c) & 1u)) {      // wave-uniform
            const char *p = (const char *)Cr.ptr[c];
            const int w = C.width(c);
            if (!VEC) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    v[c][2 * u] = load_scalar(p, w, base + (int64_t)u * (BS * 2));
                    v[c][2 * u + 1] = load_scalar(p, w, base + (int64_t)u * (BS * 2) + 1);
                }
            } else if (w == 8) {
#pragma unroll
                for (int u = 0; u < U; u++) { ll2 x = stream_load<NT, ll2>(p + (base + (int64_t)u * (BS * 2)) * 8); v[c][2 * u] = x.x; v[c][2 * u + 1] = x.y; }
            } else if (w == 4) {
#pragma unroll
                for (int u = 0; u < U; u++) { i32x2 x = stream_load<NT, i32x2>(p + (base + (int64_t)u * (BS * 2)) * 4); v[c][2 * u] = x.x; v[c][2 * u + 1] = x.y; }
            } else if (w == 2) {
#pragma unroll
                for (int u = 0; u < U; u++) { i16x2 x = stream_load<NT, i16x2>(p + (base + (int64_t)u * (BS * 2)) * 2); v[c][2 * u] = x.x; v[c][2 * u + 1] = x.y; }
            } else {
#pragma unroll
                for (int u = 0; u < U; u++) { i8x2 x = stream_load<NT, i8x2>(p + (base + (int64_t)u * (BS * 2))); v[c][2 * u] = x.x; v[c][2 * u + 1] = x.y; }
            }
        }
    }
}

template <int NC, int RW>
__device__ __forceinline__ void eval_pass(const MsArgs &C, const MScanDesc &D, const int64_t (&v)[NC][RW], bool (&pass)[RW]) {
#pragma unroll
    for (int r = 0; r < RW; r++) pass[r] = true;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if ((C.filtered >> c) & 1u) {                      // wave-uniform (bits only below ncol)
            const int64_t lo = D.flo[c], hi = D.fhi[c];
#pragma unroll
            for (int r = 0; r < RW; r++) pass[r] = pass[r] & (v[c][r] >= lo) & (v[c][r] <= hi);
        }
    }
}

// Derived columns (vdl_fuse.h VColKind): values looked up through an earlier column -- the dimension side of an FK join
// seen from the fact table (Vlite.hs:1199-1282).  `alive` starts as "the direct range filters pass", so rows a cheap
// filter already rejects do no lookups (Q14 keeps 1 row in 84); a lookup out of range makes the row EPS (alive = false).
template <int NC, int RW>
__device__ __forceinline__ void derive(const MsArgs &C, const MsArgs &Cr, const MScanDesc &D, const MScanDesc &Dr, int64_t (&v)[NC][RW], bool (&alive)[RW], uint32_t only,
                                       const int64_t (&rowid)[RW] /* global row ids: Cr.row0 + index */, bool direct_filters = true) {
    if (direct_filters) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (((C.filtered >> c) & 1u) && !((C.derived >> c) & 1u)) {
                const int64_t lo = D.flo[c], hi = D.fhi[c];
#pragma unroll
                for (int r = 0; r < RW; r++) alive[r] = alive[r] & (v[c][r] >= lo) & (v[c][r] <= hi);
            }
        }
    }
    // (a row-id column has no source and may come FIRST -- the select pass of a front whose only deciding columns are the row id and
    // the set it is looked up in; every other derived column has an earlier one to derive from, so the loop starts at 1)
    if ((only & 1u) && D.dkind[0] == VC_ROWID) {
#pragma unroll
        for (int r = 0; r < RW; r++) v[0][r] = rowid[r] - Cr.rowid_base;
    }
#pragma unroll
    for (int c = 1; c < NC; c++) {
        if ((only >> c) & 1u) {                            // wave-uniform
            const int kind = D.dkind[c], a = D.dsrc[c], b = D.dsrc2[c];
            if (kind == VC_ROWID) {
#pragma unroll
                for (int r = 0; r < RW; r++) v[c][r] = rowid[r] - Cr.rowid_base;
                continue;
            }
            if (kind == VC_FORM) {
                // A boolean formula over range tests of earlier columns.  Descriptor layout (bind_forms, vdl_engine.cpp): D.dtests[c]
                // tests sorted by column -- so each column's tests run with the column index a compile-time constant, no
                // chain of selects -- then the postfix program over their result bits (REF j pushes test j), evaluated on a
                // stack of bits (bit 0 = top).  The tests of a row slice in which no lane is still alive are skipped
                // (wave-uniform; its value is never read -- for Q19, where the cheap filters and the part bitmap leave 1 row
                // in 400, that is 5 slices in 6).  An instruction budget matters here: at 8 TB/s a wave has about 140 vector
                // instructions per row slice of a 28 B/row scan.
                const int L = D.dtests[c];
                bool live[RW];
                uint64_t bits[RW];
                uint32_t stk[RW];
#pragma unroll
                for (int r = 0; r < RW; r++) { live[r] = __ballot(alive[r]) != 0; bits[r] = 0; stk[r] = 0; }
                VDL_SPEC_UNROLL
                for (int j = 0; j < L; j++) {              // (runtime loops outside, the unrolled ones inside: the column array stays in registers)
                    const int col = D.form[a + j].col;
                    const int64_t lo = D.form[a + j].lo, hi = D.form[a + j].hi;
#pragma unroll
                    for (int k = 0; k < NC; k++) {
                        if (k < c && k == col) {           // scalar branch: one body runs
#pragma unroll
                            for (int r = 0; r < RW; r++)
                                if (live[r]) bits[r] |= (uint64_t)((v[k][r] >= lo) & (v[k][r] <= hi)) << j;
                        }
                    }
                }
                VDL_SPEC_UNROLL
                for (int s = a + L; s < a + b; s++) {
                    const int op = D.form[s].op;
                    if (op == FormStep::REF) {
                        const int j = D.form[s].col;
#pragma unroll
                        for (int r = 0; r < RW; r++) stk[r] = (stk[r] << 1) | (uint32_t)((bits[r] >> j) & 1ull);
                    } else if (op == FormStep::AND) {
#pragma unroll
                        for (int r = 0; r < RW; r++) stk[r] = ((stk[r] >> 1) & ~1u) | (stk[r] & (stk[r] >> 1) & 1u);
                    } else if (op == FormStep::OR) {
#pragma unroll
                        for (int r = 0; r < RW; r++) stk[r] = (stk[r] >> 1) | (stk[r] & 1u);
                    } else if (op == FormStep::NOT) {
#pragma unroll
                        for (int r = 0; r < RW; r++) stk[r] ^= 1u;
                    } else {
#pragma unroll
                        for (int r = 0; r < RW; r++) stk[r] = (stk[r] << 1) | (op == FormStep::TRUE_ ? 1u : 0u);
                    }
                }
#pragma unroll
                for (int r = 0; r < RW; r++) v[c][r] = (int64_t)(stk[r] & 1u);
                if ((C.filtered >> c) & 1u) {
                    const int64_t lo = D.flo[c], hi = D.fhi[c];
#pragma unroll
                    for (int r = 0; r < RW; r++) alive[r] = alive[r] & (v[c][r] >= lo) & (v[c][r] <= hi);
                }
                continue;
            }
            int64_t x[RW], y[RW];
#pragma unroll
            for (int r = 0; r < RW; r++) { x[r] = 0; y[r] = 0; }
#pragma unroll
            for (int k = 0; k < NC; k++) {                 // register files are not indexable: select the source column by comparison
                if (k < c && k == a) {
#pragma unroll
                    for (int r = 0; r < RW; r++) x[r] = v[k][r];
                }
                if (k < c && k == b) {
#pragma unroll
                    for (int r = 0; r < RW; r++) y[r] = v[k][r];
                }
            }
            if (kind == VC_SUB) {
#pragma unroll
                for (int r = 0; r < RW; r++) v[c][r] = (int64_t)((uint64_t)x[r] - (uint64_t)y[r]);
                if ((C.filtered >> c) & 1u) {               // (the projection scan has no second look at the filters: fold it here)
                    const int64_t lo = D.flo[c], hi = D.fhi[c];
#pragma unroll
                    for (int r = 0; r < RW; r++) alive[r] = alive[r] & (v[c][r] >= lo) & (v[c][r] <= hi);
                }
                continue;
            }
            const int64_t n = Dr.dn[c];
            const char *t = (const char *)Cr.ptr[c];
            const int w = C.width(c);
            bool in[RW];
#pragma unroll
            for (int r = 0; r < RW; r++) in[r] = alive[r] & (x[r] >= 0) & (x[r] < n);
            // lookups only for the lanes whose row is still alive (a selective fact filter ahead of the join -- Q14 keeps 1 row
            // in 84 -- leaves most lanes without a memory request); all the loads of a column's rows are issued before any is used
            if (kind == VC_GATHER) {
                int64_t q[RW];
#pragma unroll
                for (int r = 0; r < RW; r++) { q[r] = 0; if (in[r]) q[r] = load_scalar(t, w, x[r]); }
#pragma unroll
                for (int r = 0; r < RW; r++) { v[c][r] = q[r]; alive[r] = in[r]; }
            } else if (kind == VC_BITS) {
                uint64_t word[RW];
#pragma unroll
                for (int r = 0; r < RW; r++) word[r] = ~0ull;                      // no bitmap: every dimension row is selected
                if (t && n > 0) {                               // (unconditional: rows that are out read word 0; the bitmap is small and cached, and the loads go out together)
#pragma unroll
                    for (int r = 0; r < RW; r++) word[r] = ((const uint64_t *)t)[(in[r] ? x[r] : 0) >> 6];
                }
#pragma unroll
                for (int r = 0; r < RW; r++) { v[c][r] = in[r] ? (int64_t)((word[r] >> (x[r] & 63)) & 1ull) : 0; alive[r] = in[r]; }
            } else if (kind == VC_LUT) {                    // outside the table: 0, not EPS (Like over an offset outside the heap)
#pragma unroll
                for (int r = 0; r < RW; r++) { v[c][r] = 0; if (in[r]) v[c][r] = ((const int64_t *)t)[x[r]]; }
            } else {                                        // VC_INRANGE
#pragma unroll
                for (int r = 0; r < RW; r++) { v[c][r] = 1; alive[r] = in[r]; }
            }
            // a filter on the looked-up value (the dimension selection's bit, a dimension column's range) takes effect at
            // once: the lookups of the columns after it are then issued for the rows that are still in
            if ((C.filtered >> c) & 1u) {
                const int64_t lo = D.flo[c], hi = D.fhi[c];
#pragma unroll
                for (int r = 0; r < RW; r++) alive[r] = alive[r] & (v[c][r] >= lo) & (v[c][r] <= hi);
            }
        }
    }
}

// term of one aggregate for the lane's rows: product of affine column factors (a + s*col), a constant,
// or the row id (AGG_FIRST).  Everything read from `d` is wave-uniform (scalar loads).
template <int NC, int RW>
__device__ __forceinline__ void eval_term(const MAggDesc &d, const int64_t (&v)[NC][RW], const int64_t (&rowid)[RW], int64_t (&t)[RW]) {
    if (d.kind == AGG_FIRST) {
#pragma unroll
        for (int r = 0; r < RW; r++) t[r] = rowid[r];
        return;
    }
    const uint32_t used = d.used, plain = d.plain;
    bool first = true;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if ((used >> c) & 1u) {                            // wave-uniform
            int64_t x[RW];
            if ((plain >> c) & 1u) {
#pragma unroll
                for (int r = 0; r < RW; r++) x[r] = v[c][r];
            } else {
                const int64_t a = d.fa[c], s = d.fs[c];
                if (s == 1) {
#pragma unroll
                    for (int r = 0; r < RW; r++) x[r] = (int64_t)((uint64_t)a + (uint64_t)v[c][r]);
                } else if (s == -1) {
#pragma unroll
                    for (int r = 0; r < RW; r++) x[r] = (int64_t)((uint64_t)a - (uint64_t)v[c][r]);
                } else {
#pragma unroll
                    for (int r = 0; r < RW; r++) x[r] = (int64_t)((uint64_t)a + (uint64_t)s * (uint64_t)v[c][r]);
                }
            }
            if (first) {
#pragma unroll
                for (int r = 0; r < RW; r++) t[r] = x[r];
            } else {
#pragma unroll
                for (int r = 0; r < RW; r++) t[r] = (int64_t)((uint64_t)t[r] * (uint64_t)x[r]);
            }
            first = false;
        }
    }
    if (first) {
        const int64_t k = d.constant;
#pragma unroll
        for (int r = 0; r < RW; r++) t[r] = k;
    }
}

// LDS use: global form (1 + nagg) * 256 lane slots; grouped form replicas * (pcount * (1 + nagg) | 1) + 256 + nagg + 1 trash words
// C / D: what is known when the plan is bound (column kinds and widths, filters, key, conditions, aggregates); Cr / Dr: what
// is only known at launch (column bases, row counts, lookup-table sizes, the partials area).  The precompiled kernels pass the
// same objects for both; a kernel specialised for one plan (vdl_jit.cpp) passes compile-time constants for C and D, and the
// compiler folds every descriptor-driven branch and loop of this body away.
template <int NC, int U, bool VEC, bool NT, bool GROUPED, bool DER, bool STAGED = false>
__device__ __forceinline__ void mscan_body(const MsArgs &C, const MsArgs &Cr, const MScanDesc &D, const MScanDesc &Dr) {
    extern __shared__ int64_t lds[];
    constexpr int BS = kMsBlock, TILE = BS * 2 * U, ROWS = 2 * U;
    const int tid = threadIdx.x;
    const int nagg = D.nagg;
    const int W = nagg + 1;
    const int64_t G = D.pcount;
    const int64_t words = G * W;
    const int64_t rstride = words | 1;                     // odd int64 stride: replicas start on different banks
    const int R = D.replicas;

    int64_t cnt = 0, oob = 0;
    int64_t *mytab = lds;
    int trash = 0;
#ifdef VDL_CENSUS
    // a census build (measurement, never timed): the generated late loads count, per column, the distinct 128-byte lines they
    // ask for -- the memory side fetches whole lines (tools/ubench/fetch_calib) -- in lane 0's registers
    unsigned long long census_cnt[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) census_cnt[c] = 0;
#endif
    if (GROUPED) {
        for (int r = 0; r < R; r++)
            for (int64_t i = tid; i < words; i += BS) {
                const int w = (int)(i % W);
                lds[(int64_t)r * rstride + i] = r_identity(w == 0 ? R_SUM : rk_of(D.agg[w - 1].kind));
            }
        mytab = lds + (int64_t)(tid % R) * rstride;
        // per-lane trash rows (W words each) live behind the replicas; offset relative to mytab
        // lane t's trash row is words [t, t + W) of the trash area: rows of neighbouring lanes overlap, but within one
        // atomic instruction every lane addresses its own word (same 1 + j for all), so there is no conflict
        trash = (int)((int64_t)R * rstride + (int64_t)tid - (int64_t)(tid % R) * rstride);
        for (int64_t i = tid; i < BS + W; i += BS) lds[(int64_t)R * rstride + i] = 0;
    } else {
        for (int j = 0; j < nagg; j++) lds[(int64_t)j * BS + tid] = r_identity(rk_of(D.agg[j].kind));
    }
    __syncthreads();

    auto process = [&](auto rows_tag, int64_t (&v)[NC][decltype(rows_tag)::value], const int64_t (&rowid)[decltype(rows_tag)::value], int64_t rows_left, auto staged_tag) {
        constexpr int RW = decltype(rows_tag)::value;
        bool pass[RW];
        // Staged reads (STAGED: specialised builds, when the tuner found them quicker): only the most selective filter column comes
        // with the tile; each further filter column is read for the rows still in (masked 8-byte loads: a 64-byte sector
        // without a live row is never touched), then what derived columns and the group key need, and last the columns that
        // are only aggregate inputs, for the rows that passed everything.  Q6 keeps 1.9 % of its rows and moves ~13 of its
        // 28 B/row this way; Q14 (1 row in 84 passes the date filter) ~6 of 28.  Worthless when most rows pass (Q1).
        constexpr bool staged = STAGED && decltype(staged_tag)::value != 0;
        bool alive[RW];
#pragma unroll
        for (int r = 0; r < RW; r++) alive[r] = !DER || (int64_t)((r >> 1) * (BS * 2) + (r & 1)) < rows_left;
        // (the stages arrive as generated straight-line code -- VDL_STAGED_PRE / _POST, vdl_jit.cpp: written as loops over
        // columns and stages with the stage numbers read from C, the compiler no longer folded the descriptor: 235 KB of code)
#ifdef VDL_STAGED_PRE
        if (staged) { VDL_STAGED_PRE }
#endif
        // (scans with derived columns run their last, partial tile through this same code: rows past the end were switched off
        // above -- `rows_left` counts from the lane's first row.  Staged: the filters on table columns are already in `alive`.)
        if (DER) derive<NC, RW>(C, Cr, D, Dr, v, alive, C.derived, rowid, !staged);
        if (staged) {
#pragma unroll
            for (int r = 0; r < RW; r++) pass[r] = alive[r];
#ifdef VDL_STAGED_POST
            VDL_STAGED_POST
#endif
        } else {
            eval_pass<NC, RW>(C, D, v, pass);
            if (DER) {
#pragma unroll
                for (int r = 0; r < RW; r++) pass[r] = pass[r] & alive[r];
            }
        }
        int off[RW];
        if (GROUPED) {
            // group key: two-accumulator program (vdl_fuse.h KeyStep)
            int64_t acc[RW], tmp[RW];
#pragma unroll
            for (int r = 0; r < RW; r++) { acc[r] = 0; tmp[r] = 0; }
            const int ncomp = D.ncomp;
            if (ncomp > 0) {
                // the composite key in straight-line code: no step records, no operator dispatch (interpreting Q1's nine
                // steps was a quarter of the kernel)
#pragma unroll
                for (int k = 0; k < kMaxKeyComps; k++) {
                    if (k < ncomp) {                           // wave-uniform
                        const KeyComp kc = D.comp[k];
                        int64_t x[RW];
#pragma unroll
                        for (int r = 0; r < RW; r++) x[r] = 0;
#pragma unroll
                        for (int c = 0; c < NC; c++) {
                            if (c == kc.col) {
#pragma unroll
                                for (int r = 0; r < RW; r++) x[r] = v[c][r];
                            }
                        }
#pragma unroll
                        for (int r = 0; r < RW; r++)
                            acc[r] |= (int64_t)(((uint64_t)(x[r] >> kc.rsh) - (uint64_t)kc.sub) << kc.lsh);
                    }
                }
                if (D.key_masked) {
                    const int64_t mk = D.key_mask;
#pragma unroll
                    for (int r = 0; r < RW; r++) acc[r] &= mk;
                }
            } else
            VDL_SPEC_UNROLL
            for (int s = 0; s < D.nkey; s++) {
                const KeyStep st = D.key[s];               // wave-uniform
                if (st.kind == KeyStep::LOAD) {
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        if (c == st.col) {
                            if (st.target) {
#pragma unroll
                                for (int r = 0; r < RW; r++) tmp[r] = v[c][r];
                            } else {
#pragma unroll
                                for (int r = 0; r < RW; r++) acc[r] = v[c][r];
                            }
                        }
                    }
                } else if (st.kind == KeyStep::OPK) {
                    if (st.target) key_rows<RW>(st.bin, st.const_left, tmp, st.k);
                    else key_rows<RW>(st.bin, st.const_left, acc, st.k);
                } else {
                    key_combine<RW>(st.bin, st.const_left, acc, tmp);
                }
            }
#pragma unroll
            for (int r = 0; r < RW; r++) {
                const int64_t b = (int64_t)((uint64_t)acc[r] - (uint64_t)D.pmin);
                const bool in = b >= 0 && b < G;
                oob += (pass[r] && !in) ? 1 : 0;
                pass[r] = pass[r] && in;
                // rows that do not count go to this lane's own trash slot with the identity: no branches
                off[r] = pass[r] ? (int)b * W : trash;         // the trash row absorbs whatever is added: no selects below
                atomicAdd((unsigned long long *)&mytab[off[r]], 1ull);
            }
        } else {
#pragma unroll
            for (int r = 0; r < RW; r++) cnt += pass[r] ? 1 : 0;
        }
        VDL_SPEC_UNROLL
        for (int j = 0; j < nagg; j++) {                   // runtime loop: descriptors by scalar loads (specialised: unrolled)
            const MAggDesc d = D.agg[j];                   // whole descriptor into SGPRs: one scalar-load wait per aggregate
            const int rk = rk_of(d.kind);
            int64_t t[RW];
            eval_term<NC, RW>(d, v, rowid, t);
            if (GROUPED) {
                if (rk == R_SUM) {
#pragma unroll
                    for (int r = 0; r < RW; r++) atomicAdd((unsigned long long *)&mytab[off[r] + 1 + j], (unsigned long long)t[r]);
                } else if (rk == R_MAX) {
#pragma unroll
                    for (int r = 0; r < RW; r++) atomicMax((long long *)&mytab[off[r] + 1 + j], (long long)t[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < RW; r++) atomicMin((long long *)&mytab[off[r] + 1 + j], (long long)t[r]);
                }
            } else {
                int64_t s = r_identity(rk);
                if (rk == R_SUM) {
#pragma unroll
                    for (int r = 0; r < RW; r++) s = (int64_t)((uint64_t)s + (uint64_t)(pass[r] ? t[r] : 0));
                } else if (rk == R_MIN) {
#pragma unroll
                    for (int r = 0; r < RW; r++) s = (pass[r] && t[r] < s) ? t[r] : s;
                } else {
#pragma unroll
                    for (int r = 0; r < RW; r++) s = (pass[r] && t[r] > s) ? t[r] : s;
                }
                int64_t *slot = &lds[(int64_t)j * BS + tid];      // this lane's own slot: no atomics, no conflicts
                *slot = r_combine(rk, *slot, s);
            }
        }
    };

    const int64_t ntiles = Cr.n / TILE;
#ifdef VDL_QUEUE_FILTER
    // The QUEUE form (C.queued; specialised builds whose first filter keeps a few rows in a hundred -- Q14: one month of seven years):
    // the filter column comes with the tile and is tested there; the rows still in are queued per wave (LDS ring of row numbers), and as
    // soon as a wave holds 64 of them every lane takes one and runs the whole pipeline on it -- the other columns at the row, the
    // lookups, the aggregates.  The staged forms above do the same work in the tile's row layout, where a wave has four or five lanes
    // alive per stage and as many loads in flight: three dependent stages deep they were bound by latency at 0.41 of the peak
    // (profiles/r04/q14_bound.txt).
    if (STAGED && C.queued) {
        // (a wave holds fewer than 64 rows when it pushes a tile's -- at most 64 per row slot --, and only then takes them out 64 at a
        // time: ONE call site for the pipeline below; inlined at every push it was three copies, the lambdas stopped being inlined and
        // the descriptor went to scratch memory)
        constexpr int QNEED = kWave - 1 + ROWS * kWave;
        constexpr int QCAP = QNEED <= 128 ? 128 : QNEED <= 256 ? 256 : QNEED <= 512 ? 512 : 1024;
        static_assert(QNEED <= QCAP, "the queue holds a tile's rows beside what was left");
        __shared__ int64_t queue[kMsBlock / kWave][QCAP];
        const int lane = tid & (kWave - 1), wave = tid / kWave;
        int head = 0, held = 0;                                    // wave-uniform
        const int64_t all_tiles = (Cr.n + TILE - 1) / TILE;
        for (int64_t tile = blockIdx.x;; tile += gridDim.x) {
            const bool flush = tile >= all_tiles;                  // (block-uniform) the trip after the last tile takes out what is left
            if (!flush) {
                int64_t v[NC][ROWS], rowid[ROWS];
                const int64_t base = tile * TILE + (int64_t)tid * 2;
#pragma unroll
                for (int r = 0; r < ROWS; r++) rowid[r] = Cr.row0 + base + (int64_t)(r >> 1) * (BS * 2) + (r & 1);
                if (tile < ntiles) load_tile<NC, U, VEC, NT>(C, Cr, base, v, C.lazy);
                else {                                             // the partial last tile: clamped scalar loads
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        if (c < C.ncol && !(((C.derived | C.lazy) >> c) & 1u)) {
#pragma unroll
                            for (int r = 0; r < ROWS; r++) { const int64_t i = rowid[r] - Cr.row0; v[c][r] = load_scalar(Cr.ptr[c], C.width(c), i < Cr.n ? i : Cr.n - 1); }
                        }
                    }
                }
                bool alive[ROWS];
#pragma unroll
                for (int r = 0; r < ROWS; r++) alive[r] = rowid[r] - Cr.row0 < Cr.n;
                constexpr int RW = ROWS;
                VDL_QUEUE_FILTER
#ifdef VDL_CENSUS
                // the lines the queued rows will ask for, counted here where the rows still lie in address order: a lane's two rows are
                // adjacent, the wave's lanes hold consecutive addresses (stride 2 w), so a lane is the first asker of its 128-byte line
                // when no lower lane with a row still in lies in the same line (as the staged forms' generated census does)
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    if (c < C.ncol && ((C.lazy >> c) & 1u) && !((C.derived >> c) & 1u)) {
                        const int w = C.width(c);
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            const bool any = alive[2 * u] | alive[2 * u + 1];
                            const uint64_t m = __ballot(any);
                            const uint64_t a = (uint64_t)Cr.ptr[c] + (uint64_t)(rowid[2 * u] - Cr.row0) * (uint64_t)w;
                            int lo = lane - (int)((a & 127ull) / (uint64_t)(2 * w));
                            if (lo < 0) lo = 0;
                            const bool first = any && ((m >> lo) & ((1ull << (lane - lo)) - 1ull)) == 0ull;
                            const uint64_t f = __ballot(first);
                            if (lane == 0) census_cnt[c] += (unsigned long long)__popcll(f);
                        }
                    }
                }
#endif
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    const uint64_t m = __ballot(alive[r]);
                    if (alive[r]) queue[wave][(head + held + __popcll(m & ((1ull << lane) - 1))) & (QCAP - 1)] = rowid[r];
                    held += __popcll(m);
                }
            }
            while (held >= (flush ? 1 : kWave)) {                  // every lane takes one queued row and runs the whole pipeline on it
                const int k = held < kWave ? held : kWave;
                if (lane < k) {
                    int64_t v1[NC][1], rid[1];
                    rid[0] = queue[wave][(head + lane) & (QCAP - 1)];
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        v1[c][0] = 0;
                        if (c < C.ncol && !((C.derived >> c) & 1u)) v1[c][0] = load_scalar(Cr.ptr[c], C.width(c), rid[0] - Cr.row0);
                    }
                    process(IntTag<1>{}, v1, rid, 1, IntTag<0>{});
                }
                head = (head + k) & (QCAP - 1);
                held -= k;
            }
            if (flush) break;
        }
    } else
#endif
    {
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int64_t v[NC][ROWS], rowid[ROWS];
        const int64_t base = tile * TILE + (int64_t)tid * 2;
#pragma unroll
        for (int u = 0; u < U; u++) { rowid[2 * u] = Cr.row0 + base + (int64_t)u * (BS * 2); rowid[2 * u + 1] = rowid[2 * u] + 1; }
        load_tile<NC, U, VEC, NT>(C, Cr, base, v, STAGED ? C.lazy : 0u);
        process(IntTag<ROWS>{}, v, rowid, (int64_t)1 << 40, IntTag<1>{});
    }
    if (blockIdx.x == gridDim.x - 1 && ntiles * TILE < Cr.n) {
        if (DER) {
            // the partial tile, in the tile's own row layout with clamped scalar loads
            int64_t v[NC][ROWS], rowid[ROWS];
            const int64_t base = ntiles * TILE + (int64_t)tid * 2;
#pragma unroll
            for (int r = 0; r < ROWS; r++) rowid[r] = Cr.row0 + base + (int64_t)(r >> 1) * (BS * 2) + (r & 1);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (c < C.ncol && !((C.derived >> c) & 1u)) {
#pragma unroll
                    for (int r = 0; r < ROWS; r++) {
                        const int64_t i = base + (int64_t)(r >> 1) * (BS * 2) + (r & 1);
                        v[c][r] = load_scalar(Cr.ptr[c], C.width(c), i < Cr.n ? i : Cr.n - 1);
                    }
                }
            }
            process(IntTag<ROWS>{}, v, rowid, Cr.n - base, IntTag<1>{});
        } else {
            for (int64_t i = ntiles * TILE + tid; i < Cr.n; i += BS) {      // tail rows, one per lane
                int64_t v1[NC][1], rid[1];
                rid[0] = Cr.row0 + i;
#pragma unroll
                for (int c = 0; c < NC; c++)
                    if (c < C.ncol) v1[c][0] = load_scalar(Cr.ptr[c], C.width(c), i);
                process(IntTag<1>{}, v1, rid, 1, IntTag<1>{});
            }
        }
    }
    }
    __syncthreads();
    __shared__ int64_t red[kMsBlock / kWave];
    const int lane = tid & (kWave - 1), wave = tid / kWave;
#ifdef VDL_CENSUS
    if (lane == 0 && Dr.census) {
#pragma unroll
        for (int c = 0; c < NC; c++) if (census_cnt[c]) atomicAdd(&Dr.census[c], census_cnt[c]);
    }
#endif
    if (GROUPED) {
        int64_t *dst = Dr.block_partials + (int64_t)blockIdx.x * (words + 1);
        for (int64_t i = tid; i < words; i += BS) {
            const int w = (int)(i % W);
            const int rk = w == 0 ? R_SUM : rk_of(D.agg[w - 1].kind);
            int64_t x = lds[i];
            for (int r = 1; r < R; r++) x = r_combine(rk, x, lds[(int64_t)r * rstride + i]);
            dst[i] = x;
        }
        oob = wave_reduce(oob, R_SUM);
        if (lane == 0) red[wave] = oob;
        __syncthreads();
        if (tid == 0) { int64_t x = 0; for (int w = 0; w < kMsBlock / kWave; w++) x += red[w]; dst[words] = x; }
    } else {
        int64_t *dst = Dr.block_partials + (int64_t)blockIdx.x * W;
        for (int j = -1; j < nagg; j++) {
            const int rk = j < 0 ? R_SUM : rk_of(D.agg[j].kind);
            int64_t x = j < 0 ? cnt : lds[(int64_t)j * BS + tid];
            x = wave_reduce(x, rk);
            if (lane == 0) red[wave] = x;
            __syncthreads();
            if (tid == 0) {
                int64_t y = red[0];
                for (int w = 1; w < kMsBlock / kWave; w++) y = r_combine(rk, y, red[w]);
                dst[j + 1] = y;
            }
            __syncthreads();
        }
    }
}

// ---- projection scan (ProjPlan, vdl_fuse.h) -----------------------------------------------------------------------------
// k_project_select: ONE pass over the columns that decide a row's survival (filtered columns and what they are derived from),
// leaving the selection as a BITMAP over the table's rows (a dimension scan: Q3's orders by date and by the customer's segment,
// through the customer bitmap) or setting the bits of a semi-join set.  The fused FRONT of a fact table -- the same selection,
// then the survivors' columns as packed vectors -- is project_front_body below.  (Until round 4 the front was this pass leaving
// counts and 16-bit positions per tile, a prefix sum, and a take pass with one wave per tile.)
// Row order inside a tile is (sub-iteration u, wave, lane, row of the lane's pair).
#ifndef VDL_PROJ_U
#define VDL_PROJ_U 4                          // row pairs per lane and tile of the projection scan (variant builds: -DVDL_PROJ_U=2|8)
#endif
constexpr int kProjU = VDL_PROJ_U;
constexpr int kProjTile = kMsBlock * 2 * kProjU;
static_assert(kProjTile <= 65536, "positions inside a tile fit 16 bits");

template <int NC, int U, bool VEC, bool NT>
__device__ __forceinline__ void project_select_body(const MsArgs &C, const MsArgs &Cr, const MScanDesc &D, const MScanDesc &Dr) {
    constexpr int BS = kMsBlock, ROWS = 2 * U, TILE = BS * ROWS, NW = BS / kWave;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int64_t full = Cr.n / TILE, ntiles = (Cr.n + TILE - 1) / TILE;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int64_t v[NC][ROWS];
        const int64_t base = tile * TILE + (int64_t)tid * 2;
        if (tile < full) {
            load_tile<NC, U, VEC, NT>(C, Cr, base, v, C.lazy);
        } else {                                           // the partial last tile: clamped scalar loads
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (c < C.ncol && !(((C.derived | C.lazy) >> c) & 1u)) {
#pragma unroll
                    for (int r = 0; r < ROWS; r++) {
                        const int64_t i = base + (int64_t)(r >> 1) * (BS * 2) + (r & 1);
                        v[c][r] = load_scalar(Cr.ptr[c], C.width(c), i < Cr.n ? i : Cr.n - 1);
                    }
                }
            }
        }
        bool alive[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) alive[r] = base + (int64_t)(r >> 1) * (BS * 2) + (r & 1) < Cr.n;
        int64_t rid[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) rid[r] = Cr.row0 + base + (int64_t)(r >> 1) * (BS * 2) + (r & 1);
        derive<NC, ROWS>(C, Cr, D, Dr, v, alive, C.derived & ~C.lazy, rid);      // filters fold into `alive` as they are derived
        if (D.bitmap_only == 2) {
            // a semi-join scan: every selected row sets the bit of its index in the set (many rows share a bit: atomic OR)
            unsigned long long *set = (unsigned long long *)Dr.out_ptr[0];
            const int64_t nbits = Dr.dn[D.out_col[0]];
#pragma unroll
            for (int k = 0; k < NC; k++) {
                if (k == D.out_col[0]) {
#pragma unroll
                    for (int r = 0; r < ROWS; r++) {
                        int64_t x = v[k][r];
                        if (D.pmin > 0) x = x % D.pmin;            // the emitted `mod N` (sign of the dividend, like the element-wise operator)
                        if (alive[r] && x >= 0 && x < nbits) atomicOr(&set[x >> 6], 1ull << (x & 63));
                    }
                }
            }
            continue;
        }
        uint64_t m[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) m[r] = __ballot(alive[r]);
        // the selection's bitmap over the table's rows, for whoever asks the sparse vectors for their validity: lanes 2l, 2l+1
        // of the two ballots of a sub-iteration are rows 2l, 2l+1 of the wave's 128 -- interleave them into two words
        if (Dr.out_ptr[0] && lane < 2 * U) {
            const int u = lane >> 1, half = lane & 1;
            uint64_t a = 0, b = 0;
#pragma unroll
            for (int uu = 0; uu < U; uu++) if (uu == u) { a = m[2 * uu]; b = m[2 * uu + 1]; }
            uint64_t x = half ? (a >> 32) : (a & 0xffffffffull), y = half ? (b >> 32) : (b & 0xffffffffull);
            x = (x | (x << 16)) & 0x0000ffff0000ffffull; x = (x | (x << 8)) & 0x00ff00ff00ff00ffull; x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
            x = (x | (x << 2)) & 0x3333333333333333ull; x = (x | (x << 1)) & 0x5555555555555555ull;
            y = (y | (y << 16)) & 0x0000ffff0000ffffull; y = (y | (y << 8)) & 0x00ff00ff00ff00ffull; y = (y | (y << 4)) & 0x0f0f0f0f0f0f0f0full;
            y = (y | (y << 2)) & 0x3333333333333333ull; y = (y | (y << 1)) & 0x5555555555555555ull;
            const int64_t word = (tile * TILE + (int64_t)u * (BS * 2) + (int64_t)wave * 128) / 64 + half;
            if (word < ((Cr.n + 63) >> 6)) ((uint64_t *)Dr.out_ptr[0])[word] = x | (y << 1);
        }
    }
}

// ---- the fused front in ONE pass (round 4) ---------------------------------------------------------------------------------
// k_project_select + prefix sum + host count + k_project_take as one kernel: a block takes a tile (tiles are handed out in
// starting order), runs the select pass's work on it -- deciding columns, lookups, ballots, the survivors' ranks --, PUBLISHES the
// tile's survivor count, puts the survivors' positions (and the carried columns' values) in survivor order in LDS, learns where its
// survivors go from the counts of the tiles before it, and writes the output vectors itself: every lane takes survivors
// k, k + 256, ... as the take pass's waves did.  No tile counts, no 16-bit position scratch and no carry area in memory, no
// prefix-sum launches, no second kernel; the isolated-line fetches of the survivors' columns run beside the other blocks' streams.
// Where a tile's survivors go is a prefix over tiles that run at the same time; as in the radix Partition (vdl_partition.hip, where
// the reasoning and the measurements are) the tiles keep a Fenwick tree of fan-out 16 over their counts -- here one word per
// node {ready bit, count} --: a tile's prefix is the sum of at most 15 nodes per level, fetched by the lanes of one wave at once;
// the node of 16 / 256 / ... tiles is made by the tile that completes them.  A tile only waits for tiles that already run.
struct FrontLook {
    unsigned int *ticket;              // zeroed before the launch
    unsigned long long *nodes;         // front_look_words(tiles) words, zeroed
    int64_t *total;                    // the survivors' number, left by the last tile (device)
    int64_t *total_host;               // ... and in pinned host memory (may be null)
};
constexpr int kFrontFanBits = 4, kFrontFan = 1 << kFrontFanBits, kFrontLevels = 7;         // tiles < 16^7
constexpr unsigned long long kFrontReady = 1ull << 63;
__host__ __device__ inline int64_t front_look_words(int64_t ntiles) { int64_t w = 0; for (int j = 0; j < kFrontLevels; j++) w += ntiles >> (kFrontFanBits * j); return w; }
__device__ __forceinline__ int64_t front_node(int64_t ntiles, int64_t u, int level) {
    int64_t base = 0;
    for (int j = 0; j < level; j++) base += ntiles >> (kFrontFanBits * j);
    return base + ((u + 1) >> (kFrontFanBits * level)) - 1;
}
__device__ __forceinline__ int64_t front_wait(const unsigned long long *p) {
    unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (!(v & kFrontReady)) { __builtin_amdgcn_s_sleep(2); v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return (int64_t)(v & ~kFrontReady);
}
__device__ __forceinline__ void front_publish(unsigned long long *p, int64_t v) {
    __hip_atomic_store(p, (unsigned long long)v | kFrontReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A block takes kFrontBatch consecutive tiles at a time (a "batch"): selects them one after the other, collecting the survivors of all of
// them in LDS, and only then asks where they go and fetches their columns -- with one tile per take (the first form) a block
// waited 12 us per tile for the look-back and for the take's two rounds of dependent loads with a hundred of its 256 lanes busy, and the
// pass took 435 us for Q3 at SF10 against the two-pass front's 335 (profiles/r04/q3_front_one_tile_per_take.txt).
#ifndef VDL_FRONT_BATCH
#define VDL_FRONT_BATCH 4
#endif
constexpr int kFrontBatch = VDL_FRONT_BATCH;
static_assert(kFrontBatch * kProjTile <= 65536, "positions inside a batch fit 16 bits");
constexpr int kFrontCarry = 512;                           // carried values per batch and column; survivors beyond fetch the column again

template <int NCS, int NCT, int U, bool VEC, bool NT>
__device__ __forceinline__ void project_front_body(const MsArgs &Cs, const MsArgs &Csr, const MScanDesc &Ds, const MScanDesc &Dsr,
                                                   const MsArgs &Ct, const MsArgs &Ctr, const MScanDesc &Dt, const MScanDesc &Dtr, const FrontLook &lk) {
    constexpr int BS = kMsBlock, ROWS = 2 * U, TILE = BS * ROWS, NW = BS / kWave;
    static_assert(TILE == kProjTile, "one tile shape for the projection scans");
    __shared__ int wcnt[2][U][NW];                         // by tile parity: one barrier per tile
    __shared__ unsigned int s_batch;
    __shared__ long long s_off;
    __shared__ uint16_t spos[kFrontBatch * TILE];          // the survivors' positions inside the batch, in row order
    __shared__ int64_t cst[kMaxCarry][kFrontCarry];        // the carried columns' values, in survivor order
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int64_t full = Csr.n / TILE, ntiles = (Csr.n + TILE - 1) / TILE, nbatches = (ntiles + kFrontBatch - 1) / kFrontBatch;
    int par = 0;
    for (;;) {
        if (tid == 0) s_batch = atomicAdd(lk.ticket, 1u);
        __syncthreads();
        const int64_t batch = s_batch;
        if (batch >= nbatches) break;                      // (block-uniform)
        int total = 0;                                     // survivors of the batch so far
        for (int sub = 0; sub < kFrontBatch; sub++) {
            const int64_t tile = batch * kFrontBatch + sub;
            if (tile >= ntiles) break;                     // (block-uniform)
            int64_t v[NCS][ROWS];
            const int64_t base = tile * TILE + (int64_t)tid * 2;
            if (tile < full) {
                load_tile<NCS, U, VEC, NT>(Cs, Csr, base, v, Cs.lazy);
            } else {                                       // the partial last tile: clamped scalar loads
#pragma unroll
                for (int c = 0; c < NCS; c++) {
                    if (c < Cs.ncol && !(((Cs.derived | Cs.lazy) >> c) & 1u)) {
#pragma unroll
                        for (int r = 0; r < ROWS; r++) {
                            const int64_t i = base + (int64_t)(r >> 1) * (BS * 2) + (r & 1);
                            v[c][r] = load_scalar(Csr.ptr[c], Cs.width(c), i < Csr.n ? i : Csr.n - 1);
                        }
                    }
                }
            }
            bool alive[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) alive[r] = base + (int64_t)(r >> 1) * (BS * 2) + (r & 1) < Csr.n;
            int64_t rid[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) rid[r] = Csr.row0 + base + (int64_t)(r >> 1) * (BS * 2) + (r & 1);
            derive<NCS, ROWS>(Cs, Csr, Ds, Dsr, v, alive, Cs.derived & ~Cs.lazy, rid);      // filters fold into `alive` as they are derived
            uint64_t m[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) m[r] = __ballot(alive[r]);
            // the selection's bitmap over the table's rows, for whoever asks the sparse vectors for their validity: lanes 2l, 2l+1
            // of the two ballots of a sub-iteration are rows 2l, 2l+1 of the wave's 128 -- interleave them into two words
            if (Dsr.out_ptr[0] && lane < 2 * U) {
                const int u = lane >> 1, half = lane & 1;
                uint64_t a = 0, b = 0;
#pragma unroll
                for (int uu = 0; uu < U; uu++) if (uu == u) { a = m[2 * uu]; b = m[2 * uu + 1]; }
                uint64_t x = half ? (a >> 32) : (a & 0xffffffffull), y = half ? (b >> 32) : (b & 0xffffffffull);
                x = (x | (x << 16)) & 0x0000ffff0000ffffull; x = (x | (x << 8)) & 0x00ff00ff00ff00ffull; x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
                x = (x | (x << 2)) & 0x3333333333333333ull; x = (x | (x << 1)) & 0x5555555555555555ull;
                y = (y | (y << 16)) & 0x0000ffff0000ffffull; y = (y | (y << 8)) & 0x00ff00ff00ff00ffull; y = (y | (y << 4)) & 0x0f0f0f0f0f0f0f0full;
                y = (y | (y << 2)) & 0x3333333333333333ull; y = (y | (y << 1)) & 0x5555555555555555ull;
                const int64_t word = (tile * TILE + (int64_t)u * (BS * 2) + (int64_t)wave * 128) / 64 + half;
                if (word < ((Csr.n + 63) >> 6)) ((uint64_t *)Dsr.out_ptr[0])[word] = x | (y << 1);
            }
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < U; u++) wcnt[par][u][wave] = __popcll(m[2 * u]) + __popcll(m[2 * u + 1]);
            }
            __syncthreads();
            int here = 0, mybase[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
#pragma unroll
                for (int w = 0; w < NW; w++) { if (w == wave) mybase[u] = total + here; here += wcnt[par][u][w]; }
            }
            par ^= 1;                                      // (no second barrier: the next tile's counts go to the other half)
            const uint64_t below = (1ull << lane) - 1;
            int rank[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const int u = r >> 1;
                rank[r] = mybase[u] + __popcll(m[2 * u] & below) + __popcll(m[2 * u + 1] & below) + ((r & 1) && alive[2 * u] ? 1 : 0);
                if (alive[r]) spos[rank[r]] = (uint16_t)(sub * TILE + tid * 2 + u * (BS * 2) + (r & 1));
            }
            if (Ds.carry) {
                // the survivors' values of the deciding columns that the outputs want too (the join index of a fact table whose dimension
                // is filtered): they are in registers here -- kept in survivor order instead of being fetched again line by isolated line
                int ci = 0;
#pragma unroll
                for (int c = 0; c < NCS; c++) {
                    if ((Ds.carry >> c) & 1u) {
#pragma unroll
                        for (int r = 0; r < ROWS; r++) if (alive[r] && rank[r] < kFrontCarry) cst[ci][rank[r]] = v[c][r];
                        ci++;
                    }
                }
            }
            total += here;
        }
        // the batch's count goes out (every later batch needs it); a batch that completes 16 / 256 / ... also sums their nodes
        if (wave == 0) {
            if (lane == 0) front_publish(lk.nodes + front_node(nbatches, batch, 0), total);
            int64_t acc = total;
            for (int j = 1; j < kFrontLevels && ((batch + 1) & (((int64_t)1 << (kFrontFanBits * j)) - 1)) == 0; j++) {
                const int64_t step = (int64_t)1 << (kFrontFanBits * (j - 1));
                int64_t x = 0;
                if (lane < kFrontFan - 1) x = front_wait(lk.nodes + front_node(nbatches, batch - (int64_t)(lane + 1) * step, j - 1));
                acc += wave_reduce(x, R_SUM);              // (lane 0 holds the sums)
                if (lane == 0) front_publish(lk.nodes + front_node(nbatches, batch, j), acc);
            }
        }
        // where the survivors go: one node per unit of each hex digit of the batch number, a lane each
        if (wave == NW - 1) {
            int64_t x = 0;
            int i = lane;
            for (int j = 0; j < kFrontLevels; j++) {
                const int dgt = (int)((batch >> (kFrontFanBits * j)) & (kFrontFan - 1));
                const int64_t hi = batch & ~(((int64_t)kFrontFan << (kFrontFanBits * j)) - 1);
                for (; i < dgt; i += kWave) x += front_wait(lk.nodes + front_node(nbatches, hi + ((int64_t)(i + 1) << (kFrontFanBits * j)) - 1, j));
                i -= dgt;                                  // (lanes beyond this level's nodes move on to the next level's)
            }
            x = wave_reduce(x, R_SUM);
            if (lane == 0) s_off = x;
        }
        __syncthreads();                                   // (also: every tile's positions and carried values are in LDS)
        const int64_t off = s_off;
        if (batch == nbatches - 1 && tid == 0) {
            *lk.total = off + total;
            if (lk.total_host) __hip_atomic_store(lk.total_host, (int64_t)(off + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);    // (the host polls it)
        }
        // the survivors' rows: what the outputs need of them -- fact columns at the row (carried values from LDS), dimension columns
        // through the index -- and the packed vectors.  (Nothing is written beyond the vectors' capacity: the host may have guessed it.)
        if (off < Dtr.out_cap) {
            for (int k = tid; k < total; k += BS) {
                const int64_t row = batch * (kFrontBatch * TILE) + (int64_t)spos[k];
                int64_t vt[NCT][1];
#pragma unroll
                for (int c = 0; c < NCT; c++) {
                    vt[c][0] = 0;
                    if (c < Ct.ncol && ((Dt.take >> c) & 1u) && !((Ct.derived >> c) & 1u)) {
                        if (((Dt.carry >> c) & 1u) && k < kFrontCarry) vt[c][0] = cst[__builtin_popcount(Dt.carry & ((1u << c) - 1u))][k];
                        else vt[c][0] = load_scalar(Ctr.ptr[c], Ct.width(c), row);
                    }
                }
                bool on[1] = {true};
                int64_t rid1[1] = {Ctr.row0 + row};
                derive<NCT, 1>(Ct, Ctr, Dt, Dtr, vt, on, Ct.derived & Dt.take, rid1, false);
                if (off + k >= Dtr.out_cap) continue;
                Dtr.out_idx[off + k] = row;
                VDL_SPEC_UNROLL
                for (int o = 0; o < Dt.nout; o++) {
                    const int oc = Dt.out_col[o];
                    int64_t x = 0;
                    if (oc >= 0) {
#pragma unroll
                        for (int c = 0; c < NCT; c++) if (c == oc) x = vt[c][0];
                    } else {
                        // a row expression over the columns, in the two-accumulator program form of the group keys (ProjPlan::exprs)
                        int64_t acc[1] = {0}, tmp[1] = {0};
                        const int at = Dt.expr_at[-2 - oc], len = Dt.expr_len[-2 - oc];
                        VDL_SPEC_UNROLL
                        for (int s = at; s < at + len; s++) {
                            const KeyStep st = Dt.key[s];               // wave-uniform
                            if (st.kind == KeyStep::LOAD) {
#pragma unroll
                                for (int c = 0; c < NCT; c++) if (c == st.col) { if (st.target) tmp[0] = vt[c][0]; else acc[0] = vt[c][0]; }
                            } else if (st.kind == KeyStep::OPK) {
                                if (st.target) key_rows<1>(st.bin, st.const_left, tmp, st.k);
                                else key_rows<1>(st.bin, st.const_left, acc, st.k);
                            } else {
                                key_combine<1>(st.bin, st.const_left, acc, tmp);
                            }
                        }
                        x = acc[0];
                    }
                    Dtr.out_ptr[o][off + k] = x;
                }
            }
        }
        __syncthreads();                                   // (the next batch reuses the LDS areas)
    }
}

}  // namespace
}  // namespace vdl

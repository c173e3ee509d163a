// vdl_ops.hip -- the per-operator kernels of the general path (any VDL program, one kernel per statement):
// element-wise operators, expression chains, column filters, selection, gathers and scatters, compaction, folds,
// cross products, LIKE.  HBM-bound integer work: batched unconditional loads, one ballot per 64 slots.
#include "vdl_device.h"

namespace vdl {

// ------------------------------------------------------------------------------------------
// per-operator kernels (the general path: any VDL program, one kernel per statement)
// ------------------------------------------------------------------------------------------
// element-wise binary (/root/reference/src/Vdl.hs:110-122,436-439): values only; validity is the
// AND of the operand bitmaps (k_and_words), so EPS slots are computed and ignored.
template <int OP>
__global__ __launch_bounds__(256) void k_binary(Src a, Src b, int64_t *__restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        int64_t x0 = ld(a, i), x1 = ld(a, i + stride), x2 = ld(a, i + 2 * stride), x3 = ld(a, i + 3 * stride);
        int64_t y0 = ld(b, i), y1 = ld(b, i + stride), y2 = ld(b, i + 2 * stride), y3 = ld(b, i + 3 * stride);
        out[i] = apply_bin(OP, x0, y0);
        out[i + stride] = apply_bin(OP, x1, y1);
        out[i + 2 * stride] = apply_bin(OP, x2, y2);
        out[i + 3 * stride] = apply_bin(OP, x3, y3);
    }
    for (; i < n; i += stride) out[i] = apply_bin(OP, ld(a, i), ld(b, i));
}

hipError_t launch_binary(int op, Src a, Src b, int64_t *out, int64_t n, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    if (n <= 0) return hipSuccess;
    const int block = 256, grid = grid_for(n, block, 4);
#define VDL_BIN(OP) case OP: k_binary<OP><<<grid, block, 0, s>>>(a, b, out, n); break;
    switch (op) {
        VDL_BIN(B_LAND) VDL_BIN(B_LOR) VDL_BIN(B_BAND) VDL_BIN(B_BOR) VDL_BIN(B_SHIFT) VDL_BIN(B_EQ)
        VDL_BIN(B_ADD) VDL_BIN(B_SUB) VDL_BIN(B_GT) VDL_BIN(B_MUL) VDL_BIN(B_DIV) VDL_BIN(B_MOD)
    default: return hipErrorInvalidValue;
    }
#undef VDL_BIN
    return launch_status();
}

// Fused element-wise expression tree.  The postfix program is wave-uniform, so the operand stack lives in
// registers with compile-time indices: every instruction switches (scalar branches) on the stack height it runs
// at and on its operator.  Each lane evaluates kExprRows rows at once to amortise the scalar work.
// The stack is D x R registers whether or not the program fills it, and the registers decide how many waves a SIMD holds: a program of
// stack depth 2 -- TPC-H Q18's group key (l_orderkey - 1) & mask, one stored leaf -- ran in the depth-8 kernel with FOUR loads per lane
// in flight and three waves per SIMD: 2.7 TB/s (700 us for 120 M rows).  Three shapes now, chosen by the program's depth:
// 2 x 8 rows, 4 x 4 rows, 8 x 4 rows.
// (the leaf's representation is resolved once per push and its loads are unconditional -- rows past n read slot 0 and are
// never stored -- so the loads of a push go out together)
#define VDL_EX_PUSH(K) case K: if constexpr (K < D) by_kind(lf.kind, [&](auto kk) { _Pragma("unroll") for (int r = 0; r < R; r++) st[K][r] = ldk<decltype(kk)::value>(lf, row[r] < n ? row[r] : 0); }); break;
// a (x) b for the lane's rows; the operator switch is wave-uniform and sits outside the row loop
#define VDL_EX_OP(OP) case OP: _Pragma("unroll") for (int r = 0; r < R; r++) a[r] = apply_bin(OP, a[r], b[r]); break;
template <int R>
__device__ __forceinline__ void expr_rows(int op, int64_t (&a)[R], const int64_t (&b)[R]) {
    switch (op) {
        VDL_EX_OP(B_LAND) VDL_EX_OP(B_LOR) VDL_EX_OP(B_BAND) VDL_EX_OP(B_BOR) VDL_EX_OP(B_SHIFT) VDL_EX_OP(B_EQ)
        VDL_EX_OP(B_ADD) VDL_EX_OP(B_SUB) VDL_EX_OP(B_GT) VDL_EX_OP(B_MUL)
        case X_GE: _Pragma("unroll") for (int r = 0; r < R; r++) a[r] = a[r] >= b[r]; break;
        case X_NE: _Pragma("unroll") for (int r = 0; r < R; r++) a[r] = a[r] != b[r]; break;
        // Divide / Modulo are kept out of fused trees (vdl_engine.cpp expr_binary): the 64-bit division routine inlined
        // at every stack height tripled the size of the kernel
    }
}
#undef VDL_EX_OP
#define VDL_EX_BIN(K) case K: if constexpr (K <= D) expr_rows<R>(op, st[K - 2], st[K - 1]); break;
template <int D, int R>
__global__ __launch_bounds__(256) void k_expr(const ExprProg P, int64_t *__restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; base < n; base += R * stride) {
        int64_t row[R];
#pragma unroll
        for (int r = 0; r < R; r++) row[r] = base + r * stride;
        int64_t st[D][R];
        int sp = 0;
        for (int k = 0; k < P.n_instr; k++) {
            const int code = P.code[k];                 // wave-uniform
            if (code < 0) {
                const Src lf = P.leaf[-code - 1];
                switch (sp) { VDL_EX_PUSH(0) VDL_EX_PUSH(1) VDL_EX_PUSH(2) VDL_EX_PUSH(3) VDL_EX_PUSH(4) VDL_EX_PUSH(5) VDL_EX_PUSH(6) VDL_EX_PUSH(7) }
                sp++;
            } else {
                const int op = code;
                switch (sp) { VDL_EX_BIN(2) VDL_EX_BIN(3) VDL_EX_BIN(4) VDL_EX_BIN(5) VDL_EX_BIN(6) VDL_EX_BIN(7) VDL_EX_BIN(8) }
                sp--;
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) if (row[r] < n) out[row[r]] = st[0][r];
    }
}
#undef VDL_EX_PUSH
#undef VDL_EX_BIN
hipError_t launch_expr(const ExprProg &prog, int64_t *out, int64_t n, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    int depth = 0, sp = 0;                                      // the stack height the program reaches
    for (int k = 0; k < prog.n_instr; k++) { sp += prog.code[k] < 0 ? 1 : -1; depth = sp > depth ? sp : depth; }
    if (depth > kExprDepth || sp != 1) return hipErrorInvalidValue;
    if (depth <= 2) k_expr<2, 8><<<grid_for(n, 256, 8), 256, 0, s>>>(prog, out, n);
    else if (depth <= 4) k_expr<4, 4><<<grid_for(n, 256, 4), 256, 0, s>>>(prog, out, n);
    else k_expr<kExprDepth, 4><<<grid_for(n, 256, 4), 256, 0, s>>>(prog, out, n);
    return launch_status();
}

// Select over unfiltered table columns (Vlite.hs:721-730) when the predicate is a conjunction of per-column interval
// sets: one pass over the columns, one ballot per 64 rows, instead of a kernel per comparison / connective.
constexpr int kFilterUnroll = 4;                       // words (of 64 rows) per wave iteration: that many loads per column in flight
// NC columns, NI intervals per column (unused intervals are empty: lo > hi), no data-independent branches in the loop
template <int NC, int NI>
__global__ __launch_bounds__(256) void k_filter_columns(const FilterArgs A, uint64_t *out, int64_t n) {
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave);
    for (int64_t w0 = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; w0 < nw; w0 += kFilterUnroll * wstride) {
        int64_t v[NC][kFilterUnroll];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            by_kind(A.col[c].kind, [&](auto k) {        // one dispatch per column, its loads back to back
#pragma unroll
                for (int u = 0; u < kFilterUnroll; u++) {
                    const int64_t i = ((w0 + u * wstride) << 6) + lane;
                    v[c][u] = ldk<decltype(k)::value>(A.col[c], i < n ? i : 0);
                }
            });
        }
#pragma unroll
        for (int u = 0; u < kFilterUnroll; u++) {
            const int64_t w = w0 + u * wstride;
            const int64_t i = (w << 6) + lane;
            bool ok = i < n && !A.never;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                bool in = false;
#pragma unroll
                for (int k = 0; k < NI; k++) in = in | ((v[c][u] >= A.lo[c][k]) & (v[c][u] <= A.hi[c][k]));
                ok = ok & in;
            }
            const uint64_t m = __ballot(ok);
            if (lane == 0 && w < nw) out[w] = m;
        }
    }
}
// The commonest first-level filter -- one 4-byte column against one interval (a date range) -- with 16-byte loads: a lane
// takes four consecutive rows, a wave 256 rows = four bitmap words, assembled from the lanes' 4-bit results by an OR
// over each group of 16 lanes.  (With one row per lane the loads are 256 B per wave instruction and the kernel reached
// 3.8 TB/s; the per-row arithmetic was never the limit.)
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int kWideGroups = 4;                          // 256-row groups a wave has in flight
__global__ __launch_bounds__(256) void k_filter_i32_wide(const int32_t *col, int64_t lo, int64_t hi, uint64_t *out, int64_t ngroups) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t gstride = (int64_t)gridDim.x * (blockDim.x / kWave) * kWideGroups;
    for (int64_t g0 = wave_index() * kWideGroups; g0 < ngroups; g0 += gstride) {
        i32x4 v[kWideGroups];
#pragma unroll
        for (int u = 0; u < kWideGroups; u++) {
            const int64_t g = g0 + u < ngroups ? g0 + u : ngroups - 1;
            v[u] = __builtin_nontemporal_load((const i32x4 *)(col + g * 256) + lane);
        }
#pragma unroll
        for (int u = 0; u < kWideGroups; u++) {
            unsigned nib = 0;
            nib |= ((int64_t)v[u].x >= lo && (int64_t)v[u].x <= hi) ? 1u : 0u;
            nib |= ((int64_t)v[u].y >= lo && (int64_t)v[u].y <= hi) ? 2u : 0u;
            nib |= ((int64_t)v[u].z >= lo && (int64_t)v[u].z <= hi) ? 4u : 0u;
            nib |= ((int64_t)v[u].w >= lo && (int64_t)v[u].w <= hi) ? 8u : 0u;
            uint64_t part = (uint64_t)nib << (4 * (lane & 15));
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) part |= __shfl_xor(part, off, kWave);
            if ((lane & 15) == 0 && g0 + u < ngroups) out[(g0 + u) * 4 + (lane >> 4)] = part;
        }
    }
}

hipError_t launch_filter_columns(const FilterArgs &a0, uint64_t *out, int64_t n, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    FilterArgs a = a0;
    int ni = 1;
    for (int c = 0; c < a.ncol; c++) ni = a.nint[c] > ni ? a.nint[c] : ni;
    ni = ni <= 1 ? 1 : ni <= 2 ? 2 : kMaxFilterIvs;
    for (int c = 0; c < a.ncol; c++)
        for (int k = a.nint[c]; k < kMaxFilterIvs; k++) { a.lo[c][k] = 1; a.hi[c][k] = 0; }       // empty
    if (a.ncol == 1 && ni == 1 && !a.never && a.col[0].kind == SRC_I32 && ((uintptr_t)a.col[0].p & 15) == 0 && n >= 4096 &&
        !getenv("VDL_NO_WIDE_FILTER")) {
        const int64_t ngroups = n / 256;
        k_filter_i32_wide<<<grid_for(ngroups, 4, kWideGroups), 256, 0, s>>>((const int32_t *)a.col[0].p, a.lo[0][0], a.hi[0][0], out, ngroups);
        const int64_t done = ngroups * 256;
        if (done == n) return launch_status();
        a.col[0].p = (const int32_t *)a.col[0].p + done;       // the last rows (fewer than 256) one per lane
        out += done / 64;
        n -= done;
    }
    const int grid = grid_for((n + 63) >> 6, 4, kFilterUnroll);
#define VDL_FC(NC, NI) k_filter_columns<NC, NI><<<grid, 256, 0, s>>>(a, out, n)
#define VDL_FN(NC) if (ni == 1) VDL_FC(NC, 1); else if (ni == 2) VDL_FC(NC, 2); else VDL_FC(NC, kMaxFilterIvs)
    switch (a.ncol) {
    case 1: VDL_FN(1); break;
    case 2: VDL_FN(2); break;
    case 3: VDL_FN(3); break;
    case 4: VDL_FN(4); break;
    case 5: VDL_FN(5); break;
    case 6: VDL_FN(6); break;
    default: return hipErrorInvalidValue;
    }
#undef VDL_FN
#undef VDL_FC
    return launch_status();
}

// Filter predicates on masks (PredProg, vdl_kernels.h).  A wave takes U bitmap words per trip.  Its mask stack lives
// across the lanes of one 64-bit register: lane depth*U + u holds the mask of word u at that stack depth, so pushing is
// a lane-id compare, a connective is one shuffle by U lanes, and nothing is indexed dynamically.
constexpr int kPredWords = 4;
static_assert(kPredDepth * kPredWords <= kWave, "the mask stack fits the lanes of a wave");
__global__ __launch_bounds__(256) void k_pred(const PredProg P, const uint64_t *valid, uint64_t *out, int64_t n) {
    constexpr int U = kPredWords;
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave) * U;
    for (int64_t w0 = wave_index() * U; w0 < nw; w0 += wstride) {
        int64_t row[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int64_t i = ((w0 + u) << 6) + lane; row[u] = i < n ? i : 0; }    // bits past n are cut below
        uint64_t stk = 0;
        int sp = 0;
        for (int k = 0; k < P.n_instr; k++) {
            const int code = P.code[k];                          // wave-uniform
            if (code >= 0) {
                const Src A = P.a[code], B = P.b[code];
                const int op = P.op[code];
                uint64_t m[U];
                by_kind(A.kind, [&](auto ka) { by_kind(B.kind, [&](auto kb) {
                    int64_t x[U], y[U];
#pragma unroll
                    for (int u = 0; u < U; u++) { x[u] = ldk<decltype(ka)::value>(A, row[u]); y[u] = ldk<decltype(kb)::value>(B, row[u]); }
                    if (op == P_GT) {
#pragma unroll
                        for (int u = 0; u < U; u++) m[u] = __ballot(x[u] > y[u]);
                    } else if (op == P_EQ) {
#pragma unroll
                        for (int u = 0; u < U; u++) m[u] = __ballot(x[u] == y[u]);
                    } else if (op == P_GE) {
#pragma unroll
                        for (int u = 0; u < U; u++) m[u] = __ballot(x[u] >= y[u]);
                    } else {
#pragma unroll
                        for (int u = 0; u < U; u++) m[u] = __ballot(x[u] != y[u]);
                    }
                }); });
#pragma unroll
                for (int u = 0; u < U; u++) if (lane == sp * U + u) stk = m[u];
                sp++;
            } else {
                const uint64_t above = __shfl_down(stk, U, kWave);      // the same word one level up
                if (lane / U == sp - 2) stk = code == P_AND ? (stk & above) : (stk | above);
                sp--;
            }
        }
        // lanes 0 .. U-1 hold the result words
        if (lane < U && w0 + lane < nw) {
            const int64_t w = w0 + lane;
            const int64_t rem = n - (w << 6);
            uint64_t keep = rem < 64 ? (1ull << rem) - 1 : ~0ull;
            if (valid) keep &= valid[w];
            out[w] = stk & keep;
        }
    }
}
hipError_t launch_pred(const PredProg &prog, const uint64_t *valid, uint64_t *out, int64_t n, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    if (prog.n_instr <= 0 || prog.n_cmp <= 0 || prog.n_cmp > kPredCmps || prog.n_instr > kPredInstrs) return hipErrorInvalidValue;
    k_pred<<<grid_for((n + 63) >> 6, 4, kPredWords), 256, 0, s>>>(prog, valid, out, n);
    return launch_status();
}

// VDL_POISON=rand: every buffer the pool hands out is filled with plausible garbage -- small non-negative integers
// (in-range slot ids / positions), dense bit patterns, -1, large values, zeros -- a different mix per buffer.  Stale
// memory in a long-running process looks like that (old bitmaps, old positions), unlike a constant fill byte, which as
// an index is always out of range and as a bitmap always the same: a read of memory nobody wrote then shows as a
// wrong result instead of depending on what ran before.
__global__ void k_poison(uint64_t *p, int64_t nw, uint64_t seed) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += stride) {
        uint64_t x = seed + 0x9E3779B97F4A7C15ULL * (uint64_t)(i + 1);
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; x ^= x >> 31;
        uint64_t v;
        switch ((seed >> 3) % 5 == 4 ? x & 7 : (seed >> 3) % 5) {       // four pure kinds of buffer, one mixed
        case 0: v = x % 4096; break;                                  // small ids
        case 1: v = x; break;                                         // dense bits / huge values
        case 2: v = ~0ull; break;
        case 3: v = x % 64; break;                                    // tiny ids: collide a lot
        case 5: v = 0; break;
        case 6: v = (x & 1) ? x % 300 : ~0ull; break;
        default: v = x % 20000; break;
        }
        p[i] = v;
    }
}
hipError_t launch_poison(void *p, size_t bytes, uint64_t seed, hipStream_t s) {
    (void)hipGetLastError();
    const int64_t nw = (int64_t)(bytes / 8);
    if (nw <= 0) return hipSuccess;
    k_poison<<<grid_for(nw, 256, 4), 256, 0, s>>>((uint64_t *)p, nw, seed);
    return launch_status();
}

__global__ void k_and_words(const uint64_t *a, const uint64_t *b, uint64_t *out, int64_t nw) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += stride) out[i] = a[i] & b[i];
}
hipError_t launch_and_words(const uint64_t *a, const uint64_t *b, uint64_t *out, int64_t nw, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    if (nw <= 0) return hipSuccess;
    k_and_words<<<grid_for(nw, 256, 1), 256, 0, s>>>(a, b, out, nw);
    return launch_status();
}

__global__ void k_fill_words(uint64_t *p, uint64_t v, int64_t nw) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += stride) p[i] = v;
}
hipError_t launch_fill_words(uint64_t *p, uint64_t v, int64_t nw, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    if (nw <= 0) return hipSuccess;
    k_fill_words<<<grid_for(nw, 256, 1), 256, 0, s>>>(p, v, nw);
    return launch_status();
}

// FoldSelect with unit runs (/root/reference/src/Vlite.hs:725-727): the output values are the
// row ids themselves (a virtual range), so only the validity bitmap is produced: one 64-bit
// ballot per wave = one bitmap word.
__global__ __launch_bounds__(256) void k_select_bitmap(Src d, const uint64_t *vd, const uint64_t *vc, uint64_t *out, int64_t n) {
    constexpr int U = 8;                                        // bitmap words per wave and trip: 4 KB of int64 data in flight per wave (one word
                                                                // per trip left a cold stream at 3.7 TB/s: 130 us for 60 M rows)
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave) * U;
    by_kind(d.kind, [&](auto kd) {
        for (int64_t w0 = wave_index() * U; w0 < nw; w0 += wstride) {
            int64_t x[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t i = ((w0 + u) << 6) + lane;
                x[u] = ldk<decltype(kd)::value>(d, i < n ? i : n - 1);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t w = w0 + u;
                if (w >= nw) break;                             // wave-uniform
                const int64_t i = (w << 6) + lane;
                uint64_t m = __ballot(i < n && x[u] != 0);
                if (vd) m &= vd[w];
                if (vc) m &= vc[w];
                if (lane == 0) out[w] = m;
            }
        }
    });
}
hipError_t launch_select_bitmap(Src d, const uint64_t *vd, const uint64_t *vc, uint64_t *out, int64_t n, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    if (n <= 0) return hipSuccess;
    k_select_bitmap<<<grid_for(n, 256, 8), 256, 0, s>>>(d, vd, vc, out, n);
    return launch_status();
}

// Global (single-run) fold (/root/reference/src/Vlite.hs:337-356 with an all-equal control
// vector, Vlite.hs:636-639): two launches, per-block partials then one block.
constexpr int kFoldBlocks = 2048;
int fold_scratch_blocks() { return kFoldBlocks; }

__global__ __launch_bounds__(256) void k_fold_global(int kind, Src d, const uint64_t *vd, const uint64_t *vc, int64_t n,
                                                     int64_t *scratch) {
    constexpr int U = 8;                                        // 4 KB of data per wave in flight: a cold stream needs it
    const int rk = kind == 1 ? R_MIN : kind == 2 ? R_MAX : R_SUM;
    const int ak = kind == 4 ? R_MIN : rk;
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave) * U;
    // per lane: the fold of its data; per wave (kept uniform, lane 0 reports them): first control slot, number of data
    // slots, and for count / choose the result itself -- all three come from the bitmap words, not from the rows
    int64_t acc = r_identity(rk);
    int64_t first = INT64_MAX, cnt = 0, chosen = INT64_MAX;
    by_kind(d.kind, [&](auto kd) { by_reduction(rk, [&](auto rc) {
        constexpr int RK = decltype(rc)::value;
        for (int64_t w0 = wave_index() * U; w0 < nw; w0 += wstride) {
            int64_t x[U];
            uint64_t mc[U], md[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t w = w0 + u < nw ? w0 + u : nw - 1;
                const int64_t i = (w << 6) + lane;
                x[u] = kind < 3 ? ldk_stream<decltype(kd)::value>(d, i < n ? i : 0) : 0;     // wave-uniform test; masked lanes read slot 0
                const int64_t rem = n - (w << 6);
                mc[u] = w0 + u < nw ? (rem < 64 ? (1ull << rem) - 1 : ~0ull) : 0ull;
            }
            if (vc) {
                uint64_t t[U];
#pragma unroll
                for (int u = 0; u < U; u++) t[u] = vc[w0 + u < nw ? w0 + u : nw - 1];
#pragma unroll
                for (int u = 0; u < U; u++) mc[u] &= t[u];
            }
#pragma unroll
            for (int u = 0; u < U; u++) md[u] = mc[u];
            if (vd) {
                uint64_t t[U];
#pragma unroll
                for (int u = 0; u < U; u++) t[u] = vd[w0 + u < nw ? w0 + u : nw - 1];
#pragma unroll
                for (int u = 0; u < U; u++) md[u] &= t[u];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t wbase = (w0 + u) << 6;
                if (mc[u] && first == INT64_MAX) first = wbase + __ffsll((long long)mc[u]) - 1;      // words come in ascending order per wave
                if (md[u] && chosen == INT64_MAX) chosen = wbase + __ffsll((long long)md[u]) - 1;
                cnt += __popcll(md[u]);
                if ((md[u] >> lane) & 1ull) acc = r_combine(RK, acc, x[u]);
            }
        }
    }); });
    if (kind == 3) acc = lane == 0 ? cnt : 0;
    else if (kind == 4) acc = lane == 0 ? chosen : INT64_MAX;
    if (lane != 0) { first = INT64_MAX; cnt = 0; }
    __shared__ int64_t red[3][256 / kWave];
    acc = wave_reduce(acc, ak); first = wave_reduce(first, R_MIN); cnt = wave_reduce(cnt, R_SUM);
    if (lane == 0) { red[0][wave] = acc; red[1][wave] = first; red[2][wave] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / kWave; w++) {
            acc = r_combine(ak, acc, red[0][w]); first = r_combine(R_MIN, first, red[1][w]); cnt += red[2][w];
        }
        scratch[3 * (int64_t)blockIdx.x + 0] = acc;
        scratch[3 * (int64_t)blockIdx.x + 1] = first;
        scratch[3 * (int64_t)blockIdx.x + 2] = cnt;
    }
}

__global__ __launch_bounds__(256) void k_fold_global_finish(int kind, Src d, const int64_t *scratch, int nblocks, int64_t *result) {
    const int rk = kind == 1 ? R_MIN : kind == 2 ? R_MAX : R_SUM;
    const int ak = kind == 4 ? R_MIN : rk;
    int64_t acc = kind == 4 ? INT64_MAX : r_identity(rk), first = INT64_MAX, cnt = 0;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        acc = r_combine(ak, acc, scratch[3 * (int64_t)b]);
        first = r_combine(R_MIN, first, scratch[3 * (int64_t)b + 1]);
        cnt += scratch[3 * (int64_t)b + 2];
    }
    __shared__ int64_t red[3][256 / kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    acc = wave_reduce(acc, ak); first = wave_reduce(first, R_MIN); cnt = wave_reduce(cnt, R_SUM);
    if (lane == 0) { red[0][wave] = acc; red[1][wave] = first; red[2][wave] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / kWave; w++) {
            acc = r_combine(ak, acc, red[0][w]); first = r_combine(R_MIN, first, red[1][w]); cnt += red[2][w];
        }
        if (kind == 4) acc = cnt > 0 ? ld(d, acc) : 0;     // FoldChoose: the first datum of the run
        result[0] = acc;
        result[1] = first == INT64_MAX ? -1 : 0;     // the first (here: only) run of a vector starts at slot 0, whatever EPS slots precede its first member
        result[2] = cnt;
    }
}

hipError_t launch_fold_global(int kind, Src d, const uint64_t *vd, const uint64_t *vc, int64_t n, int64_t *scratch,
                              int64_t *result, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    int grid = grid_for(n, 256, 8);
    if (grid > kFoldBlocks) grid = kFoldBlocks;
    k_fold_global<<<grid, 256, 0, s>>>(kind, d, vd, vc, n, scratch);
    k_fold_global_finish<<<1, 256, 0, s>>>(kind, d, scratch, grid, result);
    return launch_status();
}

// one-hot vectors {value, slot, count}: element-wise ops between fold results
__global__ void k_onehot_binary(int op, const int64_t *a, const int64_t *b, int64_t *out) {
    const bool ok = a[2] > 0 && b[2] > 0 && a[1] == b[1] && a[1] >= 0;
    out[0] = ok ? apply_bin(op, a[0], b[0]) : 0;
    out[1] = a[1];
    out[2] = ok ? 1 : 0;
}
hipError_t launch_onehot_binary(int op, const int64_t *a, const int64_t *b, int64_t *out, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    k_onehot_binary<<<1, 1, 0, s>>>(op, a, b, out);
    return launch_status();
}
__global__ void k_onehot_const(int op, const int64_t *a, int64_t k, int const_left, int64_t *out) {
    const bool ok = a[2] > 0;
    out[0] = ok ? (const_left ? apply_bin(op, k, a[0]) : apply_bin(op, a[0], k)) : 0;
    out[1] = a[1];
    out[2] = a[2];
}
hipError_t launch_onehot_const(int op, const int64_t *a, int64_t k, int const_left, int64_t *out, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    k_onehot_const<<<1, 1, 0, s>>>(op, a, k, const_left, out);
    return launch_status();
}
// A global fold result {value, first control slot, count} of one shard as three mergeable words (value by the fold's
// own reduction with its identity when nothing was folded, first slot as a global row id by MIN, count by SUM), and back.
__global__ void k_fold_words(const int64_t *rec, int rk, int64_t row0, int64_t *out) {
    out[0] = rec[2] > 0 ? rec[0] : r_identity(rk);
    out[1] = rec[1] >= 0 ? rec[1] + row0 : INT64_MAX;
    out[2] = rec[2];
}
__global__ void k_fold_record(const int64_t *words, int64_t *rec) {
    rec[0] = words[2] > 0 ? words[0] : 0;
    rec[1] = words[1] == INT64_MAX ? -1 : 0;         // merged over all shards: the run starts at global slot 0
    rec[2] = words[2];
}
hipError_t launch_fold_words(const int64_t *rec, int reduce, int64_t row0, int64_t *out, hipStream_t s) {
    (void)hipGetLastError();
    k_fold_words<<<1, 1, 0, s>>>(rec, reduce == 1 ? R_MIN : reduce == 2 ? R_MAX : R_SUM, row0, out);
    return launch_status();
}
hipError_t launch_fold_record(const int64_t *words, int64_t *rec, hipStream_t s) {
    (void)hipGetLastError();
    k_fold_record<<<1, 1, 0, s>>>(words, rec);
    return launch_status();
}
__global__ void k_or_sets(const uint64_t *g, int world, int64_t words, uint64_t *out) {
    int64_t rows = 0;
    for (int r = 0; r < world; r++) rows += (int64_t)g[(int64_t)r * (words + 1) + words];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += stride) {
        uint64_t x = 0;
        for (int r = 0; r < world; r++) x |= g[(int64_t)r * (words + 1) + w];
        if ((w << 6) >= rows) x = 0;
        else if (rows - (w << 6) < 64) x &= (1ull << (rows - (w << 6))) - 1;
        out[w] = x;
    }
}
hipError_t launch_or_sets(const uint64_t *gathered, int world, int64_t words, uint64_t *out, hipStream_t s) {
    (void)hipGetLastError();
    if (words <= 0) return hipSuccess;
    k_or_sets<<<grid_for(words, 256, 1), 256, 0, s>>>(gathered, world, words, out);
    return launch_status();
}
__global__ void k_merge_words(const int64_t *g, int world, int64_t n_words, int64_t stride, const int32_t *ops, int64_t *out, int64_t *status_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) out[i] = merge_word(g, world, n_words, ops[i], i, stride);
    if (i == 0 && status_out) {
        int64_t st = 0, who = -1;
        for (int r = world - 1; r >= 0; r--) {
            const int64_t x = g[(int64_t)r * stride + 2 * n_words];
            if (x != 0) { st = x; who = r; }
        }
        status_out[0] = st; status_out[1] = who;
    }
}
hipError_t launch_merge_words(const int64_t *gathered, int world, int64_t n_words, int64_t stride, const int32_t *ops, int64_t *out,
                              int64_t *status_out, hipStream_t s) {
    (void)hipGetLastError();
    if (n_words <= 0 && !status_out) return hipSuccess;
    k_merge_words<<<(int)((std::max<int64_t>(n_words, 1) + 255) / 256), 256, 0, s>>>(gathered, world, n_words, stride, ops, out, status_out);
    return launch_status();
}
__global__ void k_onehot_dense(const int64_t *oh, int64_t *out, uint64_t *valid, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t slot = oh[2] > 0 ? oh[1] : -1;
    const int64_t nw = (n + 63) >> 6;
    if (out)                                                   // (null: only the validity is wanted -- RangeV over a scalar record)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (i == slot) ? oh[0] : 0;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += stride)
        valid[w] = (slot >= 0 && (slot >> 6) == w) ? (1ull << (slot & 63)) : 0ull;
}
hipError_t launch_onehot_dense(const int64_t *oh, int64_t *out, uint64_t *valid, int64_t n, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    if (n <= 0) return hipSuccess;
    k_onehot_dense<<<grid_for(out ? n : (n + 63) / 64, 256, 4), 256, 0, s>>>(oh, out, valid, n);
    return launch_status();
}

// MaterializeCompact (/root/reference/src/Vdl.hs:452-453): stream compaction with the order kept.
// count -> exclusive scan of tile counts -> write; a tile is 64 bitmap words = 4096 slots.
constexpr int kCompactWords = 64;
int64_t compact_tile() { return (int64_t)kCompactWords * 64; }

__global__ __launch_bounds__(64) void k_compact_count(const uint64_t *valid, int64_t n, int64_t *counts) {
    const int64_t nw = (n + 63) >> 6;
    const int64_t w = (int64_t)blockIdx.x * kCompactWords + threadIdx.x;
    int64_t c = 0;
    if (w < nw) {
        uint64_t m = valid ? valid[w] : ~0ull;
        const int64_t rem = n - (w << 6);
        if (rem < 64) m &= (1ull << rem) - 1;
        c = __popcll(m);
    }
    c = wave_reduce(c, R_SUM);
    if (threadIdx.x == 0) counts[blockIdx.x] = c;
}
hipError_t launch_compact_count(const uint64_t *valid, int64_t n, int64_t *counts, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    const int64_t nb = (n + compact_tile() - 1) / compact_tile();
    if (nb <= 0) return hipSuccess;
    k_compact_count<<<(int)nb, 64, 0, s>>>(valid, n, counts);
    return launch_status();
}

// single-block exclusive scan (in place); total written at [nblocks]
__global__ __launch_bounds__(1024) void k_scan_counts(int64_t *c, int64_t nb, int64_t *host_total) {
    constexpr int K = 8;                                        // consecutive entries per thread: 8192 per trip of the block
    __shared__ int64_t wsum[1024 / kWave];
    __shared__ int64_t carry;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024 * K) {
        const int64_t i0 = base + (int64_t)tid * K;
        int64_t x[K], sum = 0;
#pragma unroll
        for (int k = 0; k < K; k++) { x[k] = i0 + k < nb ? c[i0 + k] : 0; sum += x[k]; }
        int64_t incl = sum;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            int64_t y = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += y;
        }
        if (lane == kWave - 1) wsum[wave] = incl;
        __syncthreads();
        int64_t wprefix = 0;
        for (int w = 0; w < wave; w++) wprefix += wsum[w];
        int64_t run = carry + wprefix + incl - sum;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++) { if (i0 + k < nb) c[i0 + k] = run; run += x[k]; }
        if (tid == 1023) carry = run;
        __syncthreads();
    }
    if (tid == 0) { c[nb] = carry; if (host_total) *host_total = carry; }      // (pinned host memory the caller reads after its synchronise)
}
// count + scan in ONE launch for vectors of up to 64 tiles (262 144 slots).
// (Small selections -- the few thousand groups left after a HAVING, the entries behind a selective filter -- paid two launches
// of 3-4 us each for a few hundred bytes of work, eighteen times per Q18.)
__global__ __launch_bounds__(1024) void k_compact_count_scan(const uint64_t *valid, int64_t n, int64_t nb, int64_t *counts, int64_t *host_total) {
    // up to 64 tiles = 4096 words: 4 words per thread, a tile = 16 neighbouring threads; the first wave scans the tile counts
    __shared__ int tilecnt[kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int64_t nw = (n + 63) >> 6;
    if (tid < kWave) tilecnt[tid] = 0;
    __syncthreads();
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int64_t w = (int64_t)tid * 4 + k;
        if (w < nw) {
            uint64_t m = valid ? valid[w] : ~0ull;
            const int64_t rem = n - (w << 6);
            if (rem < 64) m &= (1ull << rem) - 1;
            c += __popcll(m);
        }
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) c += __shfl_xor(c, off, kWave);
    if ((lane & 15) == 0) tilecnt[tid >> 4] = c;
    __syncthreads();
    if (tid < kWave) {
        const int mine = tilecnt[tid];
        int incl = mine;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) { const int y = __shfl_up(incl, off, kWave); if (lane >= off) incl += y; }
        if (tid < nb) counts[tid] = incl - mine;
        if (tid == kWave - 1) { counts[nb] = incl; if (host_total) *host_total = incl; }
    }
}
// counts[0 .. nb) = exclusive prefix of the tile populations, counts[nb] = their total (what launch_compact_count followed by
// launch_compact_scan leave)
hipError_t launch_compact_offsets(const uint64_t *valid, int64_t n, int64_t *counts, hipStream_t s, int64_t *host_total) {
    (void)hipGetLastError();
    const int64_t nb = (n + compact_tile() - 1) / compact_tile();
    if (nb <= 0) return hipSuccess;
    if (nb <= kWave) {                                         // (64 tiles = 262 144 slots: one block, four words per thread)
        k_compact_count_scan<<<1, 1024, 0, s>>>(valid, n, nb, counts, host_total);
        return launch_status();
    }
    hipError_t e = launch_compact_count(valid, n, counts, s);
    if (e != hipSuccess) return e;
    return launch_compact_scan(counts, nb, s, host_total);
}
// `k` words of device memory into PINNED host memory, then a flag word, all with system-scope stores: the host polls the flag instead
// of sleeping in hipStreamSynchronize (whose wake-up left the GPU idle for 20-30 us per round trip of a query: verdicts, counts).
// k = 0: just the flag ("the stream has come this far").
__global__ __launch_bounds__(256) void k_post_words(const int64_t *__restrict__ src, int64_t k, int64_t *dst, int64_t *flag, int64_t seq) {
    for (int64_t i = threadIdx.x; i < k; i += blockDim.x) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
hipError_t launch_post_words(const int64_t *src, int64_t k, int64_t *pinned_dst, int64_t *pinned_flag, int64_t seq, hipStream_t s) {
    (void)hipGetLastError();
    k_post_words<<<1, 256, 0, s>>>(src, k, pinned_dst, pinned_flag, seq);
    return launch_status();
}
hipError_t launch_compact_scan(int64_t *counts, int64_t nb, hipStream_t s, int64_t *host_total) {
    (void)hipGetLastError();   // see launch_status()
    k_scan_counts<<<1, 1024, 0, s>>>(counts, nb, host_total);
    return launch_status();
}

__global__ __launch_bounds__(256) void k_compact_write(Src v, const uint64_t *valid, int64_t n, const int64_t *offsets, int64_t *out, int64_t *out_pos) {
    __shared__ int wprefix[kCompactWords];
    __shared__ uint64_t wmask[kCompactWords];
    const int64_t nw = (n + 63) >> 6;
    const int64_t w0 = (int64_t)blockIdx.x * kCompactWords;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    if (tid < kCompactWords) {
        const int64_t w = w0 + tid;
        uint64_t m = 0;
        if (w < nw) {
            m = valid ? valid[w] : ~0ull;
            const int64_t rem = n - (w << 6);
            if (rem < 64) m &= (1ull << rem) - 1;
        }
        wmask[tid] = m;
        // exclusive prefix of the 64 word counts inside the first wave (a loop on one thread cost 3 us per tile: a fifth of a
        // second per billion rows at any density)
        static_assert(kCompactWords == kWave, "one lane per word of the tile");
        const int cnt = __popcll(m);
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) { const int y = __shfl_up(incl, off, kWave); if (lane >= off) incl += y; }
        wprefix[tid] = incl - cnt;
    }
    __syncthreads();
    const int64_t base = offsets[blockIdx.x];
    constexpr int U = 4, NW = 256 / kWave;
    static_assert(kCompactWords % (U * NW) == 0, "each wave takes whole groups of U words");
    by_kind(v.kind, [&](auto kv) {
        for (int k0 = wave * U; k0 < kCompactWords; k0 += NW * U) {     // U words per trip: their loads are issued together
            int64_t x[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t i = ((w0 + k0 + u) << 6) + lane;
                x[u] = ldk<decltype(kv)::value>(v, i < n ? i : 0);       // masked lanes read slot 0 (n > 0 here)
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t m = wmask[k0 + u];                        // already cut at n
                if ((m >> lane) & 1ull) {
                    const int rank = __popcll(m & ((1ull << lane) - 1));
                    out[base + wprefix[k0 + u] + rank] = x[u];
                    if (out_pos) out_pos[base + wprefix[k0 + u] + rank] = ((w0 + k0 + u) << 6) + lane;      // ... and the slot it came from (a child selection wants both)
                }
            }
        }
    });
}
hipError_t launch_compact_write(Src v, const uint64_t *valid, int64_t n, const int64_t *offsets, int64_t *out, hipStream_t s, int64_t *out_pos) {
    (void)hipGetLastError();   // see launch_status()
    const int64_t nb = (n + compact_tile() - 1) / compact_tile();
    if (nb <= 0) return hipSuccess;
    k_compact_write<<<(int)nb, 256, 0, s>>>(v, valid, n, offsets, out, out_pos);
    return launch_status();
}

// ---- every fold of one GROUP BY in one launch, results packed per run ---------------------------------------------------------
// (the tail of a plan whose front is fused: Partition -> Scatter x k -> Fold x k over the survivors, Vlite.hs:1056-1060,1082-1098.
// Statement by statement that is a head pass, a count, a compaction per FoldChoose and fill + segmented fold + compaction per
// aggregate -- a dozen launches of a few microseconds of work each; here: one pass over the entries for all of them.)
// A block owns one compaction tile (64 head words = 4096 entries), a wave walks words of it, lane l holds entry 64 w + l.  Group of
// an entry = heads before the tile (offsets) + heads before its word inside the tile + heads at or before its lane - 1.  Values are
// folded along the lanes with a segmented shuffle scan bounded by the head word; the last lane of a segment hands the partial to
// out[group]: a plain store when the run begins and ends inside the word (nobody else adds to it), an atomic otherwise.
__device__ __forceinline__ void atomic_combine(int rk, int64_t *addr, int64_t v);
// Layout: a lane owns kGroupK = 8 CONSECUTIVE entries and folds them serially (no cross-lane traffic inside its chunk: a run that
// begins and ends there is stored at once); only what is open at the chunk's ends is combined across the lanes -- one segmented
// shuffle scan per 512 entries and fold, where a lane-per-entry layout paid one per 64 (that version was instruction-bound at
// 1.5 TB/s: 200 us for two folds over 18 M entries).  A wave covers 8 head words, a block half a compaction tile.
// NF = folds handled per trip (their loads are issued together).  vec16 bit f: fold f's data is int64 at a 16-byte aligned
// address -- four 16-byte loads per lane instead of eight scalar ones.
constexpr int kGroupK = 8, kGroupSplit = 2;
// The results of the runs that END inside a wave -- all but the wave's last one -- are staged in LDS at their group's place and stored by
// consecutive lanes (round 4): a lane storing its own groups put 64 isolated 8-byte writes into every store instruction, nine of them per
// fold and 512 entries, and the stores were 290 of the kernel's 506 us for Q3's 32 M survivors at SF100 (an ablation build without them:
// 216 us; -DVDL_GF_ABLATE=1 still builds it).
#if defined(VDL_GF_ABLATE) && VDL_GF_ABLATE == 1
#define GF_STORE(g, v) do { if ((g) == -12345) stage[wave][0] = (v); } while (0)
#else
#define GF_STORE(g, v) do { stage[wave][(g) - Bw] = (v); } while (0)
#endif
template <int NF>
__global__ __launch_bounds__(256) void k_group_fold(GroupFoldArgs a, unsigned vec16, const uint64_t *heads, int64_t m, const int64_t *offsets) {
    __shared__ int wprefix[kCompactWords];
    __shared__ uint64_t wmask[kCompactWords];
    __shared__ int64_t stage[256 / kWave][kWave * kGroupK];               // per wave: the results of its groups, by group
    constexpr int K = kGroupK, NW = 256 / kWave, WW = kWave * K / 64;          // WW = head words per wave (8)
    static_assert(kCompactWords == kGroupSplit * NW * WW, "a block's waves cover its share of the tile");
    const int64_t nw = (m + 63) >> 6;
    const int64_t tile = blockIdx.x / kGroupSplit;
    const int part = blockIdx.x % kGroupSplit;
    const int64_t w0 = tile * kCompactWords;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    if (tid < kCompactWords) {
        const int64_t w = w0 + tid;
        const uint64_t hm = w < nw ? heads[w] : 0ull;             // (bits past m are clear: launch_sorted_heads / k_seg_heads)
        wmask[tid] = hm;
        const int cnt = __popcll(hm);
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) { const int y = __shfl_up(incl, off, kWave); if (lane >= off) incl += y; }
        wprefix[tid] = incl - cnt;
    }
    __syncthreads();
    const int wfirst = part * (NW * WW) + wave * WW;              // this wave's first word inside the tile
    if (((w0 + wfirst) << 6) >= m) return;                         // wave-uniform; no barrier below
    const int64_t e0 = ((w0 + wfirst) << 6) + (int64_t)lane * K;   // my first entry
    const int nv = m - e0 >= K ? K : (m - e0 > 0 ? (int)(m - e0) : 0);      // my entries inside the vector
    const unsigned hb = (unsigned)(wmask[wfirst + (lane >> 3)] >> ((lane & 7) * 8)) & 0xffu;     // their head bits
    const int c = __popc(hb);
    int incl = c;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) { const int y = __shfl_up(incl, off, kWave); if (lane >= off) incl += y; }
    const int64_t B = offsets[tile] + wprefix[wfirst] + (incl - c);           // run heads strictly before my first entry
    const int64_t Bw = offsets[tile] + wprefix[wfirst];                       // ... before the wave's: its first group's number
    const int ngw = __shfl(incl, kWave - 1, kWave);                           // run heads inside the wave
    const uint64_t headlanes = __ballot(c != 0);
    const uint64_t below = (1ull << lane) - 1;
    // lanes whose open end my chunk continues: a segment of lanes starts behind every lane that holds a head
    const uint64_t starts = (headlanes << 1) | 1ull;
    const int seg0 = 63 - __clzll((long long)(starts & (below | (1ull << lane))));
    const bool first_head_lane = c != 0 && (headlanes & below) == 0;
    const uint64_t later = lane == kWave - 1 ? 0ull : headlanes >> (lane + 1);
    const int next_head = later ? lane + __ffsll((long long)later) : -1;      // the next lane that holds a head
    const bool open_to_end = c != 0 && !later;                                // my last run is still open when the wave ends
    const int fetch_from = next_head >= 0 ? next_head : kWave - 1;
#pragma unroll
    for (int j0 = 0; j0 < kMaxGroupFolds; j0 += NF) {          // (constant indices into the argument block: scalar loads, no scratch copy)
        if (j0 >= a.nfold) break;                              // wave-uniform
        int64_t x[NF][K];
#pragma unroll
        for (int f = 0; f < NF; f++) {
            if (j0 + f >= a.nfold) break;
            const Src d = a.data[j0 + f];
            if (a.kind[j0 + f] == 3) {
#pragma unroll
                for (int j = 0; j < K; j++) x[f][j] = 1;
            } else if (((vec16 >> (j0 + f)) & 1u) && nv == K) {
                const ll2 *q = (const ll2 *)((const int64_t *)d.p + e0);
#pragma unroll
                for (int j = 0; j < K; j += 2) { const ll2 t = q[j >> 1]; x[f][j] = t.x; x[f][j + 1] = t.y; }
            } else {
#pragma unroll
                for (int j = 0; j < K; j++) x[f][j] = ld(d, j < nv ? e0 + j : (e0 < m ? e0 : 0));
            }
        }
#pragma unroll
        for (int f = 0; f < NF; f++) {
            if (j0 + f >= a.nfold) break;
            const int kind = a.kind[j0 + f];
            int64_t *out = a.out[j0 + f];
            if (kind == 4) {                                   // FoldChoose: the first value of every run
                int64_t g = B - 1;
#pragma unroll
                for (int j = 0; j < K; j++)
                    if ((hb >> j) & 1u) { g++; GF_STORE(g, x[f][j]); }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                for (int k = lane; k < ngw; k += kWave) out[Bw + k] = stage[wave][k];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();      // (the area is the next fold's)
                continue;
            }
            const int rk = kind == 1 ? R_MIN : kind == 2 ? R_MAX : R_SUM;
            by_reduction(rk, [&](auto rc) {
                constexpr int R = decltype(rc)::value;
                int64_t acc = r_identity(R), pre = r_identity(R), g = B - 1;
                bool started = false;
#pragma unroll
                for (int j = 0; j < K; j++) {
                    if ((hb >> j) & 1u) {                      // (a head is always inside the vector)
                        if (started) GF_STORE(g, acc); else pre = acc;          // a run that began and ended in my chunk: stored at once
                        g++; acc = x[f][j]; started = true;
                    } else if (j < nv) {
                        acc = r_combine(R, acc, x[f][j]);
                    }
                }
                const int64_t tail = started ? acc : r_identity(R);         // open at my chunk's end (group g)
                int64_t S = started ? pre : acc;                            // continues what was open before my chunk (group B - 1)
#pragma unroll
                for (int off = 1; off < kWave; off <<= 1) {
                    const int64_t y = __shfl_up(S, off, kWave);
                    if (lane - off >= seg0) S = r_combine(R, S, y);
                }
                // S of a lane with a head (or of lane 63) = everything between the previous head lane and this lane's first head
                const int64_t cont = __shfl(S, fetch_from, kWave);
                if (c != 0) {
                    if (next_head >= 0) GF_STORE(g, r_combine(R, tail, cont));                                  // the run ends inside the wave
                    else atomic_combine(R, &out[g], lane == kWave - 1 ? tail : r_combine(R, tail, cont));    // it may go on in the next wave
                    if (first_head_lane && B > 0) atomic_combine(R, &out[B - 1], S);                         // ... and this one came from an earlier wave
                } else if (headlanes == 0 && lane == kWave - 1 && B > 0) {
                    atomic_combine(R, &out[B - 1], S);                                                       // a wave inside one long run
                }
            });
            // every group but the wave's last one has ended inside the wave and lies in the staging area: consecutive lanes store them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            for (int k = lane; k < ngw - 1; k += kWave) out[Bw + k] = stage[wave][k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
    }
}
hipError_t launch_group_fold(const GroupFoldArgs &a, const uint64_t *heads, int64_t m, const int64_t *offsets, hipStream_t s) {
    (void)hipGetLastError();
    const int64_t nb = (m + compact_tile() - 1) / compact_tile();
    if (nb <= 0 || a.nfold <= 0) return hipSuccess;
    if (a.nfold > kMaxGroupFolds) return hipErrorInvalidValue;
    unsigned vec16 = 0;
    for (int f = 0; f < a.nfold; f++)
        if (a.data[f].kind == SRC_I64 && ((uintptr_t)a.data[f].p & 15u) == 0) vec16 |= 1u << f;
    if (a.nfold <= 2) k_group_fold<2><<<(int)(nb * kGroupSplit), 256, 0, s>>>(a, vec16, heads, m, offsets);
    else k_group_fold<4><<<(int)(nb * kGroupSplit), 256, 0, s>>>(a, vec16, heads, m, offsets);
    return launch_status();
}

// ---- FoldSelect over general runs (never emitted by mplan2vdl, whose six call sites use unit runs: Vlite.hs:702-1228) ----
__global__ __launch_bounds__(256) void k_run_heads(const int64_t *ctl, int64_t m, int64_t *flags, int64_t *flags_copy) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += stride) {
        const int64_t f = (e == 0 || ctl[e] != ctl[e - 1]) ? 1 : 0;
        flags[e] = f; flags_copy[e] = f;
    }
}
hipError_t launch_run_heads(const int64_t *ctl, int64_t m, int64_t *flags, int64_t *flags_copy, hipStream_t s) {
    (void)hipGetLastError();
    if (m <= 0) return hipSuccess;
    k_run_heads<<<grid_for(m, 256, 4), 256, 0, s>>>(ctl, m, flags, flags_copy);
    return launch_status();
}
__global__ __launch_bounds__(256) void k_fsel_keys(const int64_t *excl_heads, const int64_t *flags, const int64_t *d, const uint64_t *vd, int64_t m,
                                                   int64_t *keys, uint64_t *selected) {
    const int64_t nw = (m + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave);
    for (int64_t w = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; w < nw; w += wstride) {
        const int64_t e = (w << 6) + lane;
        const bool sel = e < m && bit(vd, e) && d[e] != 0;
        if (e < m) keys[e] = 2 * (excl_heads[e] + flags[e] - 1) + (sel ? 0 : 1);      // run number, selected entries first
        const uint64_t mk = __ballot(sel);
        if (lane == 0) selected[w] = mk;
    }
}
hipError_t launch_fsel_keys(const int64_t *excl_heads, const int64_t *flags, const int64_t *d, const uint64_t *vd, int64_t m,
                            int64_t *keys, uint64_t *selected, hipStream_t s) {
    (void)hipGetLastError();
    if (m <= 0) return hipSuccess;
    k_fsel_keys<<<grid_for(m, 256, 4), 256, 0, s>>>(excl_heads, flags, d, vd, m, keys, selected);
    return launch_status();
}

__global__ __launch_bounds__(256) void k_select_gather(Src src, const uint64_t *vsrc, int64_t nsrc, Src pos, const uint64_t *vpos, const uint64_t *vc,
                                                       uint64_t *out, int64_t n) {
    constexpr int U = kGatherUnroll;
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave) * U;
    const int64_t w_first = wave_index() * U;
    if (nsrc <= 0) {                                        // nothing to read from: nothing is selected
        for (int64_t w = w_first + lane; w < nw; w += wstride) if (lane < U) out[w] = 0;
        return;
    }
    auto body = [&](auto kp, auto ks, auto vs) {
        for (int64_t w0 = w_first; w0 < nw; w0 += wstride) {
            int64_t pc[U];
            bool ok[U];
            gather_slots<decltype(kp)::value, decltype(vs)::value>(pos, vpos, vc, vsrc, nsrc, n, nw, w0, lane, pc, ok);
            int64_t x[U];
#pragma unroll
            for (int u = 0; u < U; u++) x[u] = ldk<decltype(ks)::value>(src, pc[u]);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t m = __ballot(ok[u] & (x[u] != 0));
                if (lane == 0 && w0 + u < nw) out[w0 + u] = m;
            }
        }
    };
    by_kind(pos.kind, [&](auto kp) {
        by_kind(src.kind, [&](auto ks) {
            if (vsrc) body(kp, ks, std::true_type{});
            else body(kp, ks, std::false_type{});
        });
    });
}
// The join filter as the compiler emits it: the fact side's 4-byte FK column against the dimension side's mask, which
// after the view scatter is a constant with a validity bitmap -- bit i of the result = row i takes part (vc) and the bit
// of dimension row fk[i] is set.  Four consecutive rows per lane (one 16-byte load of the FK column), a wave covers 256
// rows = four result words, put together by an OR over each group of 16 lanes (as in k_filter_i32_wide).
constexpr int kFkGroups = 2;                             // 256-row groups a wave has in flight in the join filter (16 lookups per lane and group)
template <typename T>
__global__ __launch_bounds__(256) void k_select_fk_bitmap(const T *fk, const uint64_t *vfk, const uint64_t *dim_bits, int64_t ndim,
                                                          const uint64_t *vc, uint64_t *out, int64_t ngroups) {
    typedef T tx2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t gstride = (int64_t)gridDim.x * (blockDim.x / kWave) * kFkGroups;
    for (int64_t g0 = wave_index() * kFkGroups; g0 < ngroups; g0 += gstride) {
        tx2 pa[kFkGroups], pb[kFkGroups];                   // rows 4l, 4l+1 and 4l+2, 4l+3 of the group
        uint64_t take[kFkGroups];
#pragma unroll
        for (int u = 0; u < kFkGroups; u++) {
            const int64_t g = g0 + u < ngroups ? g0 + u : ngroups - 1;
            const tx2 *base = (const tx2 *)(fk + g * 256) + 2 * lane;
            pa[u] = __builtin_nontemporal_load(base);
            pb[u] = __builtin_nontemporal_load(base + 1);
            take[u] = vc ? vc[g * 4 + (lane >> 4)] : ~0ull;          // the word of my 16-lane group
            if (vfk) take[u] &= vfk[g * 4 + (lane >> 4)];
        }
        uint64_t w[kFkGroups][4];
        bool in[kFkGroups][4];
#pragma unroll
        for (int u = 0; u < kFkGroups; u++) {
            const int64_t q[4] = {(int64_t)pa[u].x, (int64_t)pa[u].y, (int64_t)pb[u].x, (int64_t)pb[u].y};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                in[u][j] = q[j] >= 0 && q[j] < ndim;
                w[u][j] = dim_bits[in[u][j] ? (q[j] >> 6) : 0];       // the dimension bitmap is small and stays in cache
            }
        }
#pragma unroll
        for (int u = 0; u < kFkGroups; u++) {
            const int64_t q[4] = {(int64_t)pa[u].x, (int64_t)pa[u].y, (int64_t)pb[u].x, (int64_t)pb[u].y};
            unsigned nib = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) nib |= (in[u][j] && ((w[u][j] >> (q[j] & 63)) & 1ull)) ? (1u << j) : 0u;
            nib &= (unsigned)((take[u] >> (4 * (lane & 15))) & 15ull);
            uint64_t part = (uint64_t)nib << (4 * (lane & 15));
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) part |= __shfl_xor(part, off, kWave);
            if ((lane & 15) == 0 && g0 + u < ngroups) out[(g0 + u) * 4 + (lane >> 4)] = part;
        }
    }
}

hipError_t launch_select_gather(Src src, const uint64_t *vsrc, int64_t nsrc, Src pos, const uint64_t *vpos, const uint64_t *vc, uint64_t *out,
                                int64_t n, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    if ((pos.kind == SRC_I32 || pos.kind == SRC_I64) && src.kind == SRC_RANGE && src.step == 0 && src.from != 0 && vsrc && nsrc > 0 && n >= 4096 &&
        ((uintptr_t)pos.p & 15) == 0 && !getenv("VDL_NO_WIDE_FILTER")) {
        const int64_t ngroups = n / 256;
        const int grid = grid_for(ngroups, 4, kFkGroups);
        if (pos.kind == SRC_I32) k_select_fk_bitmap<int32_t><<<grid, 256, 0, s>>>((const int32_t *)pos.p, vpos, vsrc, nsrc, vc, out, ngroups);
        else k_select_fk_bitmap<int64_t><<<grid, 256, 0, s>>>((const int64_t *)pos.p, vpos, vsrc, nsrc, vc, out, ngroups);
        const int64_t done = ngroups * 256;
        if (done == n) return launch_status();
        pos.p = (const char *)pos.p + done * (pos.kind == SRC_I32 ? 4 : 8);          // the last rows (fewer than 256) one per lane
        if (vc) vc += done / 64;
        if (vpos) vpos += done / 64;
        out += done / 64;
        n -= done;
    }
    k_select_gather<<<grid_for(n, 256, 4), 256, 0, s>>>(src, vsrc, nsrc, pos, vpos, vc, out, n);
    return launch_status();
}

__global__ __launch_bounds__(256) void k_set_bits(const int64_t *idx, int64_t m, uint64_t *bitmap, int64_t nbits) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) {
        const int64_t i = idx[k];
        if ((uint64_t)i < (uint64_t)nbits) atomicOr((unsigned long long *)&bitmap[i >> 6], 1ull << (i & 63));
    }
}
hipError_t launch_set_bits(const int64_t *idx, int64_t m, uint64_t *bitmap, hipStream_t s, int64_t nbits) {
    (void)hipGetLastError();
    if (m <= 0) return hipSuccess;
    k_set_bits<<<grid_for(m, 256, 4), 256, 0, s>>>(idx, m, bitmap, nbits);
    return launch_status();
}

// Gather (/root/reference/src/Vdl.hs:438): out_i = src[pos_i]; EPS if pos_i is EPS / out of range /
// the source slot is EPS.  One ballot per wave writes the validity word.
__global__ __launch_bounds__(256) void k_gather(Src src, const uint64_t *vsrc, int64_t nsrc, Src pos, const uint64_t *vpos,
                                                int64_t n, int64_t *out, uint64_t *vout) {
    constexpr int U = kGatherUnroll;
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave) * U;
    const int64_t w_first = wave_index() * U;
    if (nsrc <= 0) {                                        // nothing to read from: every slot is EPS
        for (int64_t w0 = w_first; w0 < nw; w0 += wstride)
            for (int u = 0; u < U && w0 + u < nw; u++) {
                const int64_t i = ((w0 + u) << 6) + lane;
                if (i < n) out[i] = 0;
                if (lane == 0) vout[w0 + u] = 0;
            }
        return;
    }
    auto body = [&](auto kp, auto ks, auto vs) {
        for (int64_t w0 = w_first; w0 < nw; w0 += wstride) {
            int64_t pc[U];
            bool ok[U];
            gather_slots<decltype(kp)::value, decltype(vs)::value>(pos, vpos, nullptr, vsrc, nsrc, n, nw, w0, lane, pc, ok);
            int64_t x[U];
#pragma unroll
            for (int u = 0; u < U; u++) x[u] = ldk<decltype(ks)::value>(src, pc[u]);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t i = ((w0 + u) << 6) + lane;
                if (i < n) out[i] = ok[u] ? x[u] : 0;
                const uint64_t m = __ballot(ok[u]);
                if (lane == 0 && w0 + u < nw) vout[w0 + u] = m;
            }
        }
    };
    by_kind(pos.kind, [&](auto kp) {
        by_kind(src.kind, [&](auto ks) {
            if (vsrc) body(kp, ks, std::true_type{});
            else body(kp, ks, std::false_type{});
        });
    });
}
// Gather out of a SPARSE vector (values only on a selection: `entries` in selection order, `bitmap` over the n slots,
// wrank[w] = selected slots before bitmap word w): the entry of slot p is wrank[p / 64] + popcount(bits of word below p).
// No dense copy of the source is made (a join reads a filtered dimension column this way).
__global__ __launch_bounds__(256) void k_word_ranks(const uint64_t *bitmap, int64_t nw, int64_t *counts) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += stride) counts[w] = __popcll(bitmap[w]);
}
hipError_t launch_word_counts(const uint64_t *bitmap, int64_t nw, int64_t *counts, hipStream_t s) {
    (void)hipGetLastError();
    if (nw <= 0) return hipSuccess;
    k_word_ranks<<<grid_for(nw, 256, 4), 256, 0, s>>>(bitmap, nw, counts);
    return launch_status();
}
__global__ __launch_bounds__(256) void k_gather_ranked(const int64_t *entries, const uint64_t *bitmap, const int64_t *wrank, int64_t nsrc,
                                                       Src pos, const uint64_t *vpos, int64_t n, int64_t *out, uint64_t *vout) {
    constexpr int U = kGatherUnroll;
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave) * U;
    by_kind(pos.kind, [&](auto kp) {
        for (int64_t w0 = wave_index() * U; w0 < nw; w0 += wstride) {
            int64_t pc[U];
            bool ok[U];
            gather_slots<decltype(kp)::value, false>(pos, vpos, nullptr, nullptr, nsrc, n, nw, w0, lane, pc, ok);
            uint64_t word[U];
            int64_t before[U];
#pragma unroll
            for (int u = 0; u < U; u++) { word[u] = bitmap[pc[u] >> 6]; before[u] = wrank[pc[u] >> 6]; }
            int64_t e[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int b = (int)(pc[u] & 63);
                ok[u] = ok[u] & (((word[u] >> b) & 1ull) != 0);
                e[u] = ok[u] ? before[u] + __popcll(word[u] & ((1ull << b) - 1)) : 0;
            }
            int64_t x[U];
#pragma unroll
            for (int u = 0; u < U; u++) x[u] = entries[e[u]];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t i = ((w0 + u) << 6) + lane;
                if (i < n) out[i] = ok[u] ? x[u] : 0;
                const uint64_t m = __ballot(ok[u]);
                if (lane == 0 && w0 + u < nw) vout[w0 + u] = m;
            }
        }
    });
}
hipError_t launch_gather_ranked(const int64_t *entries, const uint64_t *bitmap, const int64_t *wrank, int64_t nsrc, Src pos, const uint64_t *vpos,
                                int64_t n, int64_t *out, uint64_t *vout, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    if (nsrc <= 0) return hipErrorInvalidValue;              // the caller handles empty sources
    k_gather_ranked<<<grid_for(n, 256, 4), 256, 0, s>>>(entries, bitmap, wrank, nsrc, pos, vpos, n, out, vout);
    return launch_status();
}

hipError_t launch_gather(Src src, const uint64_t *vsrc, int64_t nsrc, Src pos, const uint64_t *vpos, int64_t n, int64_t *out,
                         uint64_t *vout, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    if (n <= 0) return hipSuccess;
    k_gather<<<grid_for(n, 256, 4), 256, 0, s>>>(src, vsrc, nsrc, pos, vpos, n, out, vout);
    return launch_status();
}

// Scatter (/root/reference/src/Vdl.hs:441-442): out[pos_i] = src_i; positions are unique at every
// call site (/root/reference/src/Vlite.hs:1267,508), so plain stores do not race; the validity
// bitmap is set with 64-bit atomic OR.
__global__ __launch_bounds__(256) void k_scatter(Src src, const uint64_t *vsrc, Src pos, const uint64_t *vpos, int64_t n,
                                                 int64_t nout, int64_t *out, uint64_t *vout) {
    constexpr int U = kGatherUnroll;
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave) * U;
    const int64_t w_first = wave_index() * U;
    by_kind(pos.kind, [&](auto kp) {
        by_kind(src.kind, [&](auto ks) {
            for (int64_t w0 = w_first; w0 < nw; w0 += wstride) {
                int64_t x[U];
#pragma unroll
                for (int u = 0; u < U; u++) {                   // values and positions are loaded whether or not the slot takes part
                    const int64_t w = w0 + u < nw ? w0 + u : nw - 1;
                    const int64_t i = (w << 6) + lane;
                    x[u] = ldk<decltype(ks)::value>(src, i < n ? i : 0);
                }
                int64_t p[U];
                bool ok[U];
                gather_slots<decltype(kp)::value, false>(pos, vpos, vsrc, nullptr, nout, n, nw, w0, lane, p, ok);
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (ok[u]) {
                        out[p[u]] = x[u];
                        if (vout) atomicOr((unsigned long long *)&vout[p[u] >> 6], 1ull << (p[u] & 63));    // null: the caller knows which slots get written
                    }
                }
            }
        });
    });
}
hipError_t launch_scatter(Src src, const uint64_t *vsrc, Src pos, const uint64_t *vpos, int64_t n, int64_t nout, int64_t *out,
                          uint64_t *vout, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    if (n <= 0) return hipSuccess;
    k_scatter<<<grid_for(n, 256, 4), 256, 0, s>>>(src, vsrc, pos, vpos, n, nout, out, vout);
    return launch_status();
}


// ------------------------------------------------------------------------------------------
// Folds over a general control vector (/root/reference/src/Vlite.hs:337-356; grouped aggregates
// fold data scattered into key order, Vlite.hs:1056-1060).  Run = maximal stretch of equal control
// values, EPS control slots skipped; the result lands in the run's first slot.
//   k_seg_heads  : bitmap of run-first slots (previous non-EPS control slot found with clz on the
//                  validity words -- O(1) for the dense / prefix-valid vectors group-by produces)
//   k_seg_wordhd : per bitmap word, the last head slot at or before the word's end
//   (device-wide max-scan of that array on one block: n/64 entries)
//   k_seg_fold   : per-lane head slot, 64-lane segmented shuffle scan, one 64-bit atomic per
//                  (wave, run) onto out[head].
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t prev_valid_slot(const uint64_t *v, int64_t i) {   // nearest valid slot < i, or -1
    if (!v) return i - 1;
    int64_t w = i >> 6;
    uint64_t m = v[w] & ((1ull << (i & 63)) - 1);
    while (true) {
        if (m) return (w << 6) + 63 - __clzll((long long)m);
        if (--w < 0) return -1;
        m = v[w];
    }
}

__global__ __launch_bounds__(256) void k_seg_heads(Src ctl, const uint64_t *vc, int64_t n, uint64_t *heads) {
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave);
    for (int64_t w = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; w < nw; w += wstride) {
        const int64_t i = (w << 6) + lane;
        bool head = false;
        if (i < n && bit(vc, i)) {
            const int64_t p = prev_valid_slot(vc, i);
            head = p < 0 || ld(ctl, p) != ld(ctl, i);
        }
        const uint64_t m = __ballot(head);
        if (lane == 0) heads[w] = m;
    }
}

// The same for an int64 control vector that holds a value in every slot (what a Partition's Scatter leaves): a lane takes
// FOUR consecutive slots (two 16-byte loads, 2 KB per wave and instruction instead of 512 B), its left neighbour's last value
// comes by shuffle, and the four head bits of every lane are gathered into the wave's four bitmap words with an OR across each
// group of 16 lanes.  60 M slots: 136 -> ~80 us (the one-slot-per-lane form issued two 8-byte loads per slot).
__global__ __launch_bounds__(256) void k_seg_heads_dense(const int64_t *__restrict__ ctl, int64_t n, uint64_t *__restrict__ heads) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    constexpr int R = 2;                                              // groups of 256 slots a wave has in flight (4 KB)
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t nq = (n + 255) >> 8;                                // groups of 256 slots = 4 bitmap words
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
    for (int64_t q0 = ((int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave) * R; q0 < nq; q0 += nwaves * R) {
        int64_t v[R][4], left[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int64_t q = q0 + r, i0 = (q << 8) + 4 * lane;
            if (i0 + 4 <= n) {
                const i64x2 a = *(const i64x2 *)(ctl + i0), b = *(const i64x2 *)(ctl + i0 + 2);
                v[r][0] = a.x; v[r][1] = a.y; v[r][2] = b.x; v[r][3] = b.y;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) v[r][k] = i0 + k < n ? ctl[i0 + k] : 0;
            }
            left[r] = (lane == 0 && q > 0 && q < nq) ? ctl[(q << 8) - 1] : 0;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int64_t q = q0 + r, i0 = (q << 8) + 4 * lane;
            if (q >= nq) break;                                       // wave-uniform
            const int64_t up = __shfl_up(v[r][3], 1, kWave);
            const int64_t l = lane == 0 ? left[r] : up;
            unsigned nib = 0;
            nib |= (i0 < n && (i0 == 0 || v[r][0] != l)) ? 1u : 0u;
            nib |= (i0 + 1 < n && v[r][1] != v[r][0]) ? 2u : 0u;
            nib |= (i0 + 2 < n && v[r][2] != v[r][1]) ? 4u : 0u;
            nib |= (i0 + 3 < n && v[r][3] != v[r][2]) ? 8u : 0u;
            uint64_t m = (uint64_t)nib << (4 * (lane & 15));
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) m |= __shfl_xor(m, off, kWave);
            const int64_t w = (q << 2) + (lane >> 4);
            if ((lane & 15) == 0 && w < ((n + 63) >> 6)) heads[w] = m;
        }
    }
}

// The first run of a vector starts at slot 0 (EPS control slots ahead of its first member belong to it: an ungrouped
// aggregate is read back at position 0, Vlite.hs:693-712).  k_seg_heads marks the first member; this moves that one mark
// to slot 0.  One wave; the first non-empty word is almost always word 0 (vectors scattered into key order are dense).
__global__ __launch_bounds__(64) void k_seg_first_head(uint64_t *heads, int64_t nw) {
    const int lane = threadIdx.x;
    for (int64_t w0 = 0; w0 < nw; w0 += kWave) {
        const int64_t w = w0 + lane;
        const uint64_t m = w < nw ? heads[w] : 0ull;
        const uint64_t any = __ballot(m != 0);
        if (!any) continue;
        const int first_lane = __ffsll((long long)any) - 1;
        if (lane == first_lane) heads[w] = m & (m - 1);                    // clear the lowest mark ...
        __threadfence_block();
        if (lane == first_lane) heads[0] |= 1ull;                          // ... and set it at slot 0 (same lane: ordered stores, even when w == 0)
        return;
    }
}

__global__ void k_seg_wordhd(const uint64_t *heads, int64_t nw, int64_t *wordhd) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += stride) {
        const uint64_t m = heads[w];
        wordhd[w] = m ? (w << 6) + 63 - __clzll((long long)m) : -1;
    }
}

// in-place inclusive max-scan over n int64 (n/64 entries of the head table): block maxima -> one-block
// scan of the maxima -> per-block scan with carry.  (A single-block version was 1/3 of Q3's kernel time.)
constexpr int kMxBlock = 1024, kMxItems = 4, kMxTile = kMxBlock * kMxItems;

__global__ __launch_bounds__(kMxBlock) void k_mx_block(const int64_t *x, int64_t n, int64_t *bmax) {
    __shared__ int64_t red[kMxBlock / kWave];
    const int64_t base = (int64_t)blockIdx.x * kMxTile + (int64_t)threadIdx.x * kMxItems;
    int64_t m = INT64_MIN;
#pragma unroll
    for (int k = 0; k < kMxItems; k++) if (base + k < n && x[base + k] > m) m = x[base + k];
    m = wave_reduce(m, R_MAX);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = m;
    __syncthreads();
    if (threadIdx.x == 0) { int64_t a = red[0]; for (int w = 1; w < kMxBlock / kWave; w++) a = red[w] > a ? red[w] : a; bmax[blockIdx.x] = a; }
}

// inclusive max-scan of nb block maxima on one block (nb = n / 4096)
__global__ __launch_bounds__(1024) void k_mx_scan_blocks(int64_t *x, int64_t n) {
    __shared__ int64_t wmax[1024 / kWave];
    __shared__ int64_t carry;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    if (tid == 0) carry = INT64_MIN;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + tid;
        int64_t v = i < n ? x[i] : INT64_MIN;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) { int64_t y = __shfl_up(v, off, kWave); if (lane >= off && y > v) v = y; }
        if (lane == kWave - 1) wmax[wave] = v;
        __syncthreads();
        int64_t pre = carry;
        for (int w = 0; w < wave; w++) pre = wmax[w] > pre ? wmax[w] : pre;
        v = pre > v ? pre : v;
        __syncthreads();
        if (i < n) x[i] = v;
        if (tid == 1023) carry = v;
        __syncthreads();
    }
}

__global__ __launch_bounds__(kMxBlock) void k_mx_apply(int64_t *x, int64_t n, const int64_t *bincl) {
    __shared__ int64_t wmax[kMxBlock / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int64_t base = (int64_t)blockIdx.x * kMxTile + (int64_t)tid * kMxItems;
    int64_t v[kMxItems], m = INT64_MIN;
#pragma unroll
    for (int k = 0; k < kMxItems; k++) { v[k] = (base + k < n) ? x[base + k] : INT64_MIN; if (v[k] > m) m = v[k]; }
    int64_t incl = m;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) { int64_t y = __shfl_up(incl, off, kWave); if (lane >= off && y > incl) incl = y; }
    if (lane == kWave - 1) wmax[wave] = incl;
    __syncthreads();
    int64_t run = blockIdx.x > 0 ? bincl[blockIdx.x - 1] : INT64_MIN;     // everything before this block
    for (int w = 0; w < wave; w++) run = wmax[w] > run ? wmax[w] : run;
    const int64_t prev = __shfl_up(incl, 1, kWave);                        // lanes before me in my wave
    if (lane > 0 && prev > run) run = prev;
#pragma unroll
    for (int k = 0; k < kMxItems; k++) { if (v[k] > run) run = v[k]; if (base + k < n) x[base + k] = run; }
}

int64_t maxscan_blocks(int64_t n) { return (n + kMxTile - 1) / kMxTile; }

static hipError_t launch_maxscan(int64_t *x, int64_t n, int64_t *scratch /* maxscan_blocks(n) */, hipStream_t s) {
    const int64_t nb = maxscan_blocks(n);
    if (nb <= 0) return hipSuccess;
    k_mx_block<<<(int)nb, kMxBlock, 0, s>>>(x, n, scratch);
    k_mx_scan_blocks<<<1, 1024, 0, s>>>(scratch, nb);
    k_mx_apply<<<(int)nb, kMxBlock, 0, s>>>(x, n, scratch);
    return hipGetLastError();
}

__device__ __forceinline__ void atomic_combine(int rk, int64_t *addr, int64_t v) {
    if (rk == R_SUM) atomicAdd((unsigned long long *)addr, (unsigned long long)v);
    else if (rk == R_MIN) atomicMin((long long *)addr, (long long)v);
    else atomicMax((long long *)addr, (long long)v);
}

// kind: 0 sum, 1 min, 2 max, 3 count, 4 choose-pass (min over slot index of the data)
// Each wave walks a contiguous chunk of words and carries the value of the run that is open at a word's last lane
// into the next word, so a run costs one atomic per wave it touches instead of one per 64 slots: with few long runs
// (dense GROUP BY domains) the per-word atomics all landed on the same handful of addresses and serialised.
__global__ __launch_bounds__(256) void k_seg_fold(int kind, Src d, const uint64_t *vd, const uint64_t *vc, const uint64_t *heads,
                                                  const int64_t *wordhd, int64_t n, int64_t *out, uint64_t *vout) {
    constexpr int U = kGatherUnroll;
    const int rk_rt = (kind == 1 || kind == 4) ? R_MIN : kind == 2 ? R_MAX : R_SUM;
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
    const int64_t g = wave_index();
    const int64_t per = (nw + nwaves - 1) / nwaves;
    const int64_t w_end = (g + 1) * per < nw ? (g + 1) * per : nw;
    const uint64_t upto = lane == 63 ? ~0ull : ((2ull << lane) - 1);        // lanes at or before mine
    by_kind(d.kind, [&](auto kd) { by_reduction(rk_rt, [&](auto rc) {
        constexpr int rk = decltype(rc)::value;
        int64_t carry_h = -1, carry_x = r_identity(rk);       // wave-uniform: the run still open after the previous word
        for (int64_t w0 = g * per; w0 < w_end; w0 += U) {
            // A long run (a dense-domain GROUP BY has a handful over millions of slots): while the next 16 (or 8) words begin no
            // run, hold a value in every slot and continue the run this wave carries, their 1024 (512) values are simply reduced --
            // lanes 0..15 look one word each, then every lane loads 16 (8) consecutive values with 16-byte loads: 8 (4) KB per wave
            // in flight instead of the 2 KB of the word-by-word path below, whose lane = slot layout the segments need
            // (60 M rows in 32 runs: 175 -> ~110 us).
            if (carry_h >= 0 && (kind >= 3 || (decltype(kd)::value == SRC_I64 && ((uintptr_t)d.p & 15u) == 0))) {
                bool fast = false;
                if (lane < 16) {
                    const int64_t w = w0 + lane;
                    if (w < w_end && ((w + 1) << 6) <= n)
                        fast = heads[w] == 0 && (!vc || vc[w] == ~0ull) && (!vd || vd[w] == ~0ull) && w > 0 && wordhd[w - 1] == carry_h;
                }
                const unsigned fm = (unsigned)(__ballot(fast) & 0xFFFFull);
                const int K = fm == 0xFFFFu ? 16 : (fm & 0xFFu) == 0xFFu ? 8 : 0;
                if (K) {
                    int64_t t = r_identity(rk);
                    if (kind < 3) {
                        typedef long long i64x2 __attribute__((ext_vector_type(2)));
                        const i64x2 *base = (const i64x2 *)((const int64_t *)d.p + (w0 << 6)) + lane;
                        i64x2 v[8];
#pragma unroll
                        for (int j = 0; j < 8; j++) if (j < K / 2) v[j] = base[j * kWave];
#pragma unroll
                        for (int j = 0; j < 8; j++) if (j < K / 2) t = r_combine(rk, t, r_combine(rk, v[j].x, v[j].y));
                        t = __shfl(wave_reduce(t, rk), 0, kWave);
                    } else t = kind == 3 ? (int64_t)K * 64 : (w0 << 6);
                    carry_x = r_combine(rk, carry_x, t);
                    w0 += K - U;
                    continue;
                }
            }
            // operands of U words first (the words of a wave are processed in order: the carried run links them)
            uint64_t hw_[U], okw_[U];
            int64_t x_[U], prev_[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t w = w0 + u < w_end ? w0 + u : w_end - 1;
                const int64_t i = (w << 6) + lane;
                hw_[u] = heads[w];
                prev_[u] = w > 0 ? wordhd[w - 1] : -1;
                const int64_t rem = n - (w << 6);
                okw_[u] = rem < 64 ? (1ull << rem) - 1 : ~0ull;
                x_[u] = kind < 3 ? ldk<decltype(kd)::value>(d, i < n ? i : 0) : (kind == 3 ? 1 : i);
            }
            if (vc) {
                uint64_t t[U];
#pragma unroll
                for (int u = 0; u < U; u++) t[u] = vc[w0 + u < w_end ? w0 + u : w_end - 1];
#pragma unroll
                for (int u = 0; u < U; u++) okw_[u] &= t[u];
            }
            if (vd) {
                uint64_t t[U];
#pragma unroll
                for (int u = 0; u < U; u++) t[u] = vd[w0 + u < w_end ? w0 + u : w_end - 1];
#pragma unroll
                for (int u = 0; u < U; u++) okw_[u] &= t[u];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (w0 + u >= w_end) break;                    // wave-uniform
                const int64_t w = w0 + u;
                const uint64_t hw = hw_[u];
                if (hw == 0 && okw_[u] == ~0ull && carry_h >= 0 && prev_[u] == carry_h) {
                    // no run begins in this word and all 64 slots take part: the word continues the run the wave carries (long
                    // runs -- a dense-domain GROUP BY has a handful -- are nearly all such words): one wave reduction, no segments
                    carry_x = r_combine(rk, carry_x, __shfl(wave_reduce(x_[u], rk), 0, kWave));
                    continue;
                }
                const uint64_t hm = hw & upto;                 // heads at or before this lane
                const int64_t h = hm ? (w << 6) + 63 - __clzll((long long)hm) : prev_[u];
                const bool ok = ((okw_[u] >> lane) & 1ull) != 0 && h >= 0;
                int64_t x = ok ? x_[u] : r_identity(rk);
                // lane segments = maximal stretches of consecutive active lanes with one head; an EPS slot
                // inside a run splits it into several segments, each adds its part to the same out[h]
                const int64_t hp = __shfl_up(h, 1, kWave);
                const uint64_t okm = __ballot(ok);
                const bool okp = lane > 0 && ((okm >> (lane - 1)) & 1ull);
                const bool starts = lane == 0 || !ok || !okp || hp != h;
                const uint64_t sm = __ballot(starts) & upto;
                const int seg0 = 63 - __clzll((long long)sm);  // first lane of my segment (bit 0 is always set)
#pragma unroll
                for (int off = 1; off < kWave; off <<= 1) {
                    const int64_t y = __shfl_up(x, off, kWave);
                    if (lane - off >= seg0) x = r_combine(rk, x, y);
                }
                const int64_t hn = __shfl_down(h, 1, kWave);
                const bool okn = lane < kWave - 1 && ((okm >> (lane + 1)) & 1ull);
                const bool tail = ok && (lane == kWave - 1 || !okn || hn != h);
                // runs that begin and end inside this word with every slot taking part are one segment nobody else adds
                // to: a plain store and one validity update per word (a sparse GROUP BY has ~30 such runs per word,
                // and their atomics on out[] and on the same word of vout[] were most of the kernel)
                bool whole = false;
                if ((hw >> lane) & 1) {
                    const uint64_t later = lane == kWave - 1 ? 0 : hw >> (lane + 1);
                    if (later) {
                        const int q = lane + __ffsll((long long)later);       // lane of the next head (<= 63)
                        const uint64_t range = (1ull << q) - (1ull << lane);
                        whole = (okm & range) == range;
                    }
                }
                const uint64_t wholem = __ballot(whole);
                const bool mine = tail && ((wholem >> seg0) & 1);              // my segment is such a run
                // the run carried over from the previous word: continue it in this word's first segment, or write it out
                if (carry_h >= 0) {
                    const bool ok0 = (okm & 1ull) != 0;
                    const int64_t h0 = __shfl(h, 0, kWave);
                    if (ok0 && h0 == carry_h) {
                        if (tail && seg0 == 0) x = r_combine(rk, x, carry_x);
                    } else if (lane == 0) {
                        atomic_combine(rk, &out[carry_h], carry_x);
                        atomicOr((unsigned long long *)&vout[carry_h >> 6], 1ull << (carry_h & 63));
                    }
                    carry_h = -1;
                }
                if ((okm >> (kWave - 1)) & 1ull) { carry_h = __shfl(h, kWave - 1, kWave); carry_x = __shfl(x, kWave - 1, kWave); }   // lane 63 is that segment's tail
                if (mine) {
                    out[h] = x;
                } else if (tail && lane != kWave - 1) {
                    atomic_combine(rk, &out[h], x);
                    atomicOr((unsigned long long *)&vout[h >> 6], 1ull << (h & 63));
                }
                if (lane == 0 && wholem) atomicOr((unsigned long long *)&vout[w], wholem);   // runs from other words may set bits here too
            }
        }
        if (carry_h >= 0 && lane == 0) {
            atomic_combine(rk, &out[carry_h], carry_x);
            atomicOr((unsigned long long *)&vout[carry_h >> 6], 1ull << (carry_h & 63));
        }
    }); });
}

__global__ void k_seg_fill(int64_t *out, int64_t v, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = v;
}
// the reduction's identity at the run heads only: every partial of a run lands on out[head], nothing else of `out` is ever read
// (its validity bitmap holds heads only) -- filling all n slots wrote 8 B/row for nothing (a fifth of the fold over 60 M rows)
__global__ void k_seg_init_heads(const uint64_t *heads, int64_t nw, int64_t v, int64_t *out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += stride) {
        uint64_t m = heads[w];
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            out[(w << 6) + b] = v;
            m &= m - 1;
        }
    }
}

// FoldChoose second pass: out[h] currently holds the smallest data slot of the run -> its value
__global__ void k_seg_choose_fix(Src d, const uint64_t *vout, int64_t n, int64_t *out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (bit(vout, i)) out[i] = ld(d, out[i]);
}

static int seg_fold_grid(int64_t n) {            // a few thousand waves, each with a contiguous chunk of at least a few words
    const int64_t nw = (n + 63) >> 6;
    int64_t g = (nw + 15) / 16;                  // >= 4 words per wave (4 waves per block)
    if (g > 2048) g = 2048;
    return (int)(g < 1 ? 1 : g);
}

// scratch: heads bitmap (nwords) and wordhd (nwords + maxscan_blocks(nwords) int64) supplied by the caller
// run heads of a control vector (first slot of every run, EPS control slots skipped) + the per-word lookup the fold needs
hipError_t launch_fold_heads(Src ctl, const uint64_t *vc, int64_t n, uint64_t *heads, int64_t *wordhd, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    if (n <= 0) return hipSuccess;
    const int64_t nw = (n + 63) >> 6;
    if (!vc && ctl.kind == SRC_I64 && ((uintptr_t)ctl.p & 15u) == 0 && !getenv("VDL_NO_DENSE_HEADS")) k_seg_heads_dense<<<grid_for((n + 3) / 4, 256, 4), 256, 0, s>>>((const int64_t *)ctl.p, n, heads);
    else k_seg_heads<<<grid_for(n, 256, 4), 256, 0, s>>>(ctl, vc, n, heads);
    if (vc) k_seg_first_head<<<1, kWave, 0, s>>>(heads, nw);               // (without EPS slots the first member is slot 0 already)
    k_seg_wordhd<<<grid_for(nw, 256, 1), 256, 0, s>>>(heads, nw, wordhd);
    if (launch_maxscan(wordhd, nw, wordhd + nw, s) != hipSuccess) return hipGetLastError();
    return launch_status();
}
// the fold itself, over heads computed by launch_fold_heads for the same control vector (several folds share them)
hipError_t launch_fold_runs(int kind, Src d, const uint64_t *vd, const uint64_t *vc, const uint64_t *heads, const int64_t *wordhd, int64_t n,
                            int64_t *out, uint64_t *vout /* pre-zeroed */, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    const int rk = (kind == 1 || kind == 4) ? R_MIN : kind == 2 ? R_MAX : R_SUM;
    k_seg_init_heads<<<grid_for((n + 63) >> 6, 256, 1), 256, 0, s>>>(heads, (n + 63) >> 6, rk == R_SUM ? 0 : rk == R_MIN ? INT64_MAX : INT64_MIN, out);
    k_seg_fold<<<seg_fold_grid(n), 256, 0, s>>>(kind, d, vd, vc, heads, wordhd, n, out, vout);
    if (kind == 4) k_seg_choose_fix<<<grid_for(n, 256, 4), 256, 0, s>>>(d, vout, n, out);
    return launch_status();
}
hipError_t launch_fold_segmented(int kind, Src ctl, const uint64_t *vc, Src d, const uint64_t *vd, int64_t n,
                                 uint64_t *heads, int64_t *wordhd, int64_t *out, uint64_t *vout /* pre-zeroed */, hipStream_t s) {
    hipError_t e = launch_fold_heads(ctl, vc, n, heads, wordhd, s);
    if (e != hipSuccess) return e;
    return launch_fold_runs(kind, d, vd, vc, heads, wordhd, n, out, vout, s);
}


// ------------------------------------------------------------------------------------------
// CrossProductOuter / CrossProductInner (/root/reference/src/Vdl.hs:412-416, Vlite.hs:89-95,278-289;
// emitted for joins only under --crossproduct, Mplan.hs / Vlite.hs:671-680): for left of m slots and
// right of k slots, m*k slots holding the left position i / k (outer) or the right position i % k
// (inner); only the operand lengths matter, so no slot is ever EPS.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cross(int64_t n, int64_t k, int inner, int64_t *out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = inner ? i % k : i / k;
}
hipError_t launch_cross(int64_t n, int64_t k, int inner, int64_t *out, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    k_cross<<<grid_for(n, 256, 4), 256, 0, s>>>(n, k, inner, out);
    return launch_status();
}

// ------------------------------------------------------------------------------------------
// Like (/root/reference/src/Vdl.hs:244-247,444-447): data = byte offsets into the column's string
// heap (one byte per slot, strings end at a 0 byte); SQL LIKE with '%' and '_', no escape.  One
// lane per row walks its string (dictionary-like heaps are a few KB and stay in L1/L2); greedy
// match with backtracking to the last '%'.  The pattern travels by value (SGPR/constant reads).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_like(Src data, const uint64_t *vdata, int64_t n, Src heap, const uint64_t *vheap, int64_t heap_n,
                                              const LikePattern pat, int64_t *out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int plen = pat.len;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!bit(vdata, i)) { out[i] = 0; continue; }
        const int64_t off = ld(data, i);
        int64_t r = 0;
        if (off >= 0 && off < heap_n) {
            int64_t si = off, mark = 0;
            int pi = 0, star = -1;
            bool dead = false;
            for (;;) {
                const int ch = (si < heap_n && bit(vheap, si)) ? (int)(ld(heap, si) & 0xff) : 0;
                if (!ch) break;
                const int pc = pi < plen ? (int)pat.p[pi] : -1;
                if (pc == '%') { star = pi++; mark = si; }
                else if (pc == '_' || pc == ch) { si++; pi++; }
                else if (star >= 0) { pi = star + 1; si = ++mark; }
                else { dead = true; break; }
            }
            if (!dead) {
                while (pi < plen && pat.p[pi] == '%') pi++;
                r = pi == plen;
            }
        }
        out[i] = r;
    }
}
hipError_t launch_like(Src data, const uint64_t *vdata, int64_t n, Src heap, const uint64_t *vheap, int64_t heap_n, const LikePattern &pat,
                       int64_t *out, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    k_like<<<grid_for(n, 256, 1), 256, 0, s>>>(data, vdata, n, heap, vheap, heap_n, pat, out);
    return launch_status();
}

}  // namespace vdl

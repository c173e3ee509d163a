// vdl_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4, wave64).
//
// Everything on this path is HBM-bound integer work (no MFMA anywhere: there is no dense
// contraction in a VDL program).  The rules that matter: coalesced 16-byte-per-lane column
// loads, enough bytes in flight per CU, wave-level reductions (64-lane shuffles / ballots),
// LDS only for the cross-wave step, and one pass over every byte.
#include "vdl_kernels.h"

namespace vdl {

typedef long long ll2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef char i8x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;

// ------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------
enum { R_SUM = 0, R_MIN = 1, R_MAX = 2 };

__device__ __forceinline__ int64_t r_identity(int kind) {
    return kind == R_SUM ? 0 : kind == R_MIN ? INT64_MAX : INT64_MIN;
}
__device__ __forceinline__ int64_t r_combine(int kind, int64_t a, int64_t b) {
    if (kind == R_SUM) return (int64_t)((uint64_t)a + (uint64_t)b);
    if (kind == R_MIN) return a < b ? a : b;
    return a > b ? a : b;
}
__device__ __forceinline__ int64_t wave_reduce(int64_t x, int kind) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        int64_t y = __shfl_down(x, off, kWave);
        x = r_combine(kind, x, y);
    }
    return x;
}

// ------------------------------------------------------------------------------------------
// synthetic data: v(row) = add + mul * (lo + splitmix64(seed ^ col_id*PHI ^ row) % span)
// (value ranges /root/reference/tests/tpch10noorder/bounds.csv:59-79)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

template <typename T>
__global__ void k_gen_column(T *out, int64_t row0, int64_t n, uint64_t key, int64_t lo, uint64_t span, int64_t mul,
                             int64_t add) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t h = splitmix64(key ^ (uint64_t)(row0 + i));
        out[i] = (T)(int64_t)((uint64_t)add + ((uint64_t)lo + h % span) * (uint64_t)mul);
    }
}

hipError_t launch_gen_column(void *out, int elem_bytes, int64_t row0, int64_t n, uint64_t seed, uint64_t col_id,
                             int64_t lo, int64_t hi, int64_t mul, int64_t add, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const uint64_t key = seed ^ (col_id * 0x9E3779B97F4A7C15ULL);
    const uint64_t span = (uint64_t)hi - (uint64_t)lo + 1;
    const int block = 256;
    const int grid = (int)std::min<int64_t>((n + block - 1) / block, 256 * 16);
    switch (elem_bytes) {
    case 1: k_gen_column<int8_t><<<grid, block, 0, s>>>((int8_t *)out, row0, n, key, lo, span, mul, add); break;
    case 2: k_gen_column<int16_t><<<grid, block, 0, s>>>((int16_t *)out, row0, n, key, lo, span, mul, add); break;
    case 4: k_gen_column<int32_t><<<grid, block, 0, s>>>((int32_t *)out, row0, n, key, lo, span, mul, add); break;
    case 8: k_gen_column<int64_t><<<grid, block, 0, s>>>((int64_t *)out, row0, n, key, lo, span, mul, add); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Fused scan: filter (AND of per-column closed ranges) + aggregates (sum/min/max of a product
// of affine column factors) in ONE pass over the columns.
//
// Replaces, for programs of the Q6 shape, the whole chain
//   Load x4 -> Greater/Equals/LogicalOr/LogicalAnd x19 -> FoldSelect -> Gather x3 -> Multiply -> FoldSum
// (/root/reference/README.md:39-53; lowering /root/reference/src/Vlite.hs:721-730,1048-1060).
//
// Layout: a block-iteration covers TILE = 256 threads x 2 rows x U consecutive rows.  In
// sub-iteration u, lane l of the block owns rows base + u*512 + 2l, +1, so an int64 column is one
// 16-byte load per lane (1 KiB per wave instruction, fully coalesced), an int32 column one
// 8-byte load.  All NC x U loads of an iteration are issued before the first use: with U = 4 a
// wave keeps 14 KiB (Q6) in flight.  Algorithmic traffic = sum of column widths per row (Q6: 28 B).
// Per-lane accumulators -> 64-lane shuffle reduction -> LDS across the 4 waves -> one partial
// row per block; k_scan_finish folds the partial rows (deterministic order, no atomics).
// ------------------------------------------------------------------------------------------
constexpr int kScanBlock = 256;

template <int NC, int NA, int ROWS>
__device__ __forceinline__ void scan_accumulate(const ScanArgs &A, const int64_t (&v)[NC][ROWS], int64_t (&acc)[NA],
                                                int64_t &cnt) {
    bool pass[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++) pass[r] = true;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if (c < A.ncol && A.filtered[c]) {               // wave-uniform
            const int64_t lo = A.lo[c], hi = A.hi[c];
#pragma unroll
            for (int r = 0; r < ROWS; r++) pass[r] = pass[r] & (v[c][r] >= lo) & (v[c][r] <= hi);
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; r++) cnt += pass[r] ? 1 : 0;
#pragma unroll
    for (int j = 0; j < NA; j++) {
        if (j < A.nagg) {                                // wave-uniform
            int64_t t[ROWS];
            bool first = true;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (c < A.ncol && ((A.used[j] >> c) & 1u)) {   // wave-uniform
                    const bool plain = (A.plain[j] >> c) & 1u;
                    const int64_t a = A.fa[j][c], s = A.fs[j][c];
#pragma unroll
                    for (int r = 0; r < ROWS; r++) {
                        int64_t x = plain ? v[c][r] : (int64_t)((uint64_t)a + (uint64_t)s * (uint64_t)v[c][r]);
                        t[r] = first ? x : (int64_t)((uint64_t)t[r] * (uint64_t)x);
                    }
                    first = false;
                }
            }
            if (first) {
#pragma unroll
                for (int r = 0; r < ROWS; r++) t[r] = A.constant[j];
            }
            const int kind = A.kind[j];
            if (kind == AGG_SUM) {
#pragma unroll
                for (int r = 0; r < ROWS; r++) acc[j] = (int64_t)((uint64_t)acc[j] + (uint64_t)(pass[r] ? t[r] : 0));
            } else if (kind == AGG_MIN) {
#pragma unroll
                for (int r = 0; r < ROWS; r++) acc[j] = (pass[r] && t[r] < acc[j]) ? t[r] : acc[j];
            } else {
#pragma unroll
                for (int r = 0; r < ROWS; r++) acc[j] = (pass[r] && t[r] > acc[j]) ? t[r] : acc[j];
            }
        }
    }
}

__device__ __forceinline__ int64_t load_scalar(const void *p, int width, int64_t i) {
    switch (width) {
    case 8: return ((const int64_t *)p)[i];
    case 4: return ((const int32_t *)p)[i];
    case 2: return ((const int16_t *)p)[i];
    default: return ((const int8_t *)p)[i];
    }
}

template <int NC, int NA, int U, bool VEC>
__global__ __launch_bounds__(kScanBlock) void k_scan(const ScanArgs A) {
    constexpr int TILE = kScanBlock * 2 * U;
    constexpr int ROWS = 2 * U;
    const int tid = threadIdx.x;
    int64_t acc[NA];
    int64_t cnt = 0;
#pragma unroll
    for (int j = 0; j < NA; j++) acc[j] = (j < A.nagg) ? r_identity(A.kind[j]) : 0;

    const int64_t ntiles = A.n / TILE;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int64_t v[NC][ROWS];
        const int64_t base = tile * TILE + (int64_t)tid * 2;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (c < A.ncol) {                            // wave-uniform
                const char *p = (const char *)A.ptr[c];
                const int w = A.width[c];
                if (!VEC) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        v[c][2 * u] = load_scalar(p, w, base + (int64_t)u * (kScanBlock * 2));
                        v[c][2 * u + 1] = load_scalar(p, w, base + (int64_t)u * (kScanBlock * 2) + 1);
                    }
                } else if (w == 8) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        ll2 x = *(const ll2 *)(p + (base + (int64_t)u * (kScanBlock * 2)) * 8);
                        v[c][2 * u] = x.x; v[c][2 * u + 1] = x.y;
                    }
                } else if (w == 4) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        i32x2 x = *(const i32x2 *)(p + (base + (int64_t)u * (kScanBlock * 2)) * 4);
                        v[c][2 * u] = x.x; v[c][2 * u + 1] = x.y;
                    }
                } else if (w == 2) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        i16x2 x = *(const i16x2 *)(p + (base + (int64_t)u * (kScanBlock * 2)) * 2);
                        v[c][2 * u] = x.x; v[c][2 * u + 1] = x.y;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        i8x2 x = *(const i8x2 *)(p + (base + (int64_t)u * (kScanBlock * 2)));
                        v[c][2 * u] = x.x; v[c][2 * u + 1] = x.y;
                    }
                }
            }
        }
        scan_accumulate<NC, NA, ROWS>(A, v, acc, cnt);
    }
    // tail rows (fewer than one tile) go to the last block, one row per lane
    if (blockIdx.x == gridDim.x - 1) {
        for (int64_t i = ntiles * TILE + tid; i < A.n; i += kScanBlock) {
            int64_t v1[NC][1];
#pragma unroll
            for (int c = 0; c < NC; c++)
                if (c < A.ncol) v1[c][0] = load_scalar(A.ptr[c], A.width[c], i);
            scan_accumulate<NC, NA, 1>(A, v1, acc, cnt);
        }
    }
    // block reduction: shuffles inside each wave, LDS across the 4 waves
    __shared__ int64_t red[kScanBlock / kWave][NA + 1];
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    int64_t c = wave_reduce(cnt, R_SUM);
    if (lane == 0) red[wave][0] = c;
#pragma unroll
    for (int j = 0; j < NA; j++) {
        if (j < A.nagg) {
            int64_t x = wave_reduce(acc[j], A.kind[j]);
            if (lane == 0) red[wave][j + 1] = x;
        }
    }
    __syncthreads();
    if (tid == 0) {
        int64_t *dst = A.block_partials + (int64_t)blockIdx.x * (A.nagg + 1);
        int64_t x = red[0][0];
#pragma unroll
        for (int w = 1; w < kScanBlock / kWave; w++) x += red[w][0];
        dst[0] = x;
#pragma unroll
        for (int j = 0; j < NA; j++) {
            if (j < A.nagg) {
                int64_t y = red[0][j + 1];
#pragma unroll
                for (int w = 1; w < kScanBlock / kWave; w++) y = r_combine(A.kind[j], y, red[w][j + 1]);
                dst[j + 1] = y;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_scan_finish(const int64_t *partials, int nblocks, const ScanArgs A, int64_t *words) {
    __shared__ int64_t red[256 / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
#pragma unroll
    for (int w = 0; w <= kMaxScanAggs; w++) {
        if (w > A.nagg) break;
        const int kind = w == 0 ? R_SUM : A.kind[w == 0 ? 0 : w - 1];
        int64_t x = r_identity(kind);
        for (int b = tid; b < nblocks; b += 256) x = r_combine(kind, x, partials[(int64_t)b * (A.nagg + 1) + w]);
        x = wave_reduce(x, kind);
        if (lane == 0) red[wave] = x;
        __syncthreads();
        if (tid == 0) {
            int64_t y = red[0];
            for (int k = 1; k < 256 / kWave; k++) y = r_combine(kind, y, red[k]);
            words[w] = y;
        }
        __syncthreads();
    }
}

namespace {
typedef void (*scan_fn)(const ScanArgs);
struct ScanVariant { int nc, na, u; bool vec; scan_fn fn; const char *name; };
const ScanVariant kScanVariants[] = {
    {4, 1, 4, true, k_scan<4, 1, 4, true>, "k_scan<4,1,4,vec>"},
    {4, 4, 4, true, k_scan<4, 4, 4, true>, "k_scan<4,4,4,vec>"},
    {8, 8, 2, true, k_scan<8, 8, 2, true>, "k_scan<8,8,2,vec>"},
    {4, 1, 4, false, k_scan<4, 1, 4, false>, "k_scan<4,1,4,scalar>"},
    {4, 4, 4, false, k_scan<4, 4, 4, false>, "k_scan<4,4,4,scalar>"},
    {8, 8, 2, false, k_scan<8, 8, 2, false>, "k_scan<8,8,2,scalar>"},
};
constexpr int kNumScanVariants = sizeof(kScanVariants) / sizeof(kScanVariants[0]);
}  // namespace

ScanLaunch scan_launch_config(const ScanArgs &a, int num_cus) {
    // vector loads need every column base aligned to its two-row access
    bool vec = true;
    for (int c = 0; c < a.ncol; c++)
        if (((uintptr_t)a.ptr[c]) % (uintptr_t)(2 * a.width[c]) != 0) vec = false;
    ScanLaunch cfg;
    cfg.variant = -1;
    for (int i = 0; i < kNumScanVariants; i++) {
        const ScanVariant &v = kScanVariants[i];
        if (v.vec == vec && a.ncol <= v.nc && a.nagg <= v.na) { cfg.variant = i; break; }
    }
    if (cfg.variant < 0) return cfg;
    const ScanVariant &v = kScanVariants[cfg.variant];
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v.fn, kScanBlock, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    if (per_cu > 8) per_cu = 8;
    const int64_t tile = (int64_t)kScanBlock * 2 * v.u;
    const int64_t ntiles = a.n / tile;
    int64_t grid = (int64_t)num_cus * per_cu;
    if (grid > ntiles) grid = ntiles;
    if (grid < 1) grid = 1;
    cfg.grid = (int)grid;
    cfg.block = kScanBlock;
    return cfg;
}

const char *scan_kernel_name(const ScanLaunch &cfg) {
    return (cfg.variant >= 0 && cfg.variant < kNumScanVariants) ? kScanVariants[cfg.variant].name : "none";
}

hipError_t launch_scan(const ScanArgs &a, const ScanLaunch &cfg, hipStream_t s) {
    if (cfg.variant < 0 || cfg.variant >= kNumScanVariants) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kScanVariants[cfg.variant].fn, dim3(cfg.grid), dim3(cfg.block), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_scan_finish(const int64_t *block_partials, int nblocks, int, const int *, const ScanArgs &a, int64_t *words,
                              hipStream_t s) {
    k_scan_finish<<<1, 256, 0, s>>>(block_partials, nblocks, a, words);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// per-operator kernels (the general path: any VDL program, one kernel per statement)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t ld(const Src &s, int64_t i) {
    switch (s.kind) {                                     // wave-uniform
    case SRC_I64: return ((const int64_t *)s.p)[i];
    case SRC_I32: return ((const int32_t *)s.p)[i];
    case SRC_I16: return ((const int16_t *)s.p)[i];
    case SRC_I8: return ((const int8_t *)s.p)[i];
    default: return (int64_t)((uint64_t)s.from + (uint64_t)i * (uint64_t)s.step);
    }
}
__device__ __forceinline__ bool bit(const uint64_t *v, int64_t i) { return v ? ((v[i >> 6] >> (i & 63)) & 1ull) : true; }

static inline int grid_for(int64_t n, int block, int per_thread) {
    int64_t g = (n + (int64_t)block * per_thread - 1) / ((int64_t)block * per_thread);
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

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
    if (n <= 0) return hipSuccess;
    const int block = 256, grid = grid_for(n, block, 4);
#define VDL_BIN(OP) case OP: k_binary<OP><<<grid, block, 0, s>>>(a, b, out, n); break;
    switch (op) {
        VDL_BIN(B_LAND) VDL_BIN(B_LOR) VDL_BIN(B_BAND) VDL_BIN(B_BOR) VDL_BIN(B_SHIFT) VDL_BIN(B_EQ)
        VDL_BIN(B_ADD) VDL_BIN(B_SUB) VDL_BIN(B_GT) VDL_BIN(B_MUL) VDL_BIN(B_DIV) VDL_BIN(B_MOD)
    default: return hipErrorInvalidValue;
    }
#undef VDL_BIN
    return hipGetLastError();
}

__global__ void k_and_words(const uint64_t *a, const uint64_t *b, uint64_t *out, int64_t nw) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += stride) out[i] = a[i] & b[i];
}
hipError_t launch_and_words(const uint64_t *a, const uint64_t *b, uint64_t *out, int64_t nw, hipStream_t s) {
    if (nw <= 0) return hipSuccess;
    k_and_words<<<grid_for(nw, 256, 1), 256, 0, s>>>(a, b, out, nw);
    return hipGetLastError();
}

__global__ void k_fill_words(uint64_t *p, uint64_t v, int64_t nw) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += stride) p[i] = v;
}
hipError_t launch_fill_words(uint64_t *p, uint64_t v, int64_t nw, hipStream_t s) {
    if (nw <= 0) return hipSuccess;
    k_fill_words<<<grid_for(nw, 256, 1), 256, 0, s>>>(p, v, nw);
    return hipGetLastError();
}

// FoldSelect with unit runs (/root/reference/src/Vlite.hs:725-727): the output values are the
// row ids themselves (a virtual range), so only the validity bitmap is produced: one 64-bit
// ballot per wave = one bitmap word.
__global__ __launch_bounds__(256) void k_select_bitmap(Src d, const uint64_t *vd, const uint64_t *vc, uint64_t *out, int64_t n) {
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave);
    for (int64_t w = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; w < nw; w += wstride) {
        const int64_t i = (w << 6) + lane;
        bool nz = (i < n) && (ld(d, i) != 0);
        uint64_t m = __ballot(nz);
        if (vd) m &= vd[w];
        if (vc) m &= vc[w];
        if (lane == 0) out[w] = m;
    }
}
hipError_t launch_select_bitmap(Src d, const uint64_t *vd, const uint64_t *vc, uint64_t *out, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    k_select_bitmap<<<grid_for(n, 256, 4), 256, 0, s>>>(d, vd, vc, out, n);
    return hipGetLastError();
}

// Global (single-run) fold (/root/reference/src/Vlite.hs:337-356 with an all-equal control
// vector, Vlite.hs:636-639): two launches, per-block partials then one block.
constexpr int kFoldBlocks = 2048;
int fold_scratch_blocks() { return kFoldBlocks; }

__global__ __launch_bounds__(256) void k_fold_global(int kind, Src d, const uint64_t *vd, const uint64_t *vc, int64_t n,
                                                     int64_t *scratch) {
    const int rk = kind == 1 ? R_MIN : kind == 2 ? R_MAX : R_SUM;
    int64_t acc = kind == 4 ? INT64_MAX : r_identity(rk);   // choose: smallest slot index holding a datum
    int64_t first = INT64_MAX, cnt = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const bool c_ok = bit(vc, i);
        if (c_ok && i < first) first = i;
        if (c_ok && bit(vd, i)) {
            cnt++;
            if (kind == 3) acc += 1;
            else if (kind == 4) acc = i < acc ? i : acc;
            else acc = r_combine(rk, acc, ld(d, i));
        }
    }
    __shared__ int64_t red[3][256 / kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int ak = kind == 4 ? R_MIN : rk;
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
        result[1] = first == INT64_MAX ? -1 : first;
        result[2] = cnt;
    }
}

hipError_t launch_fold_global(int kind, Src d, const uint64_t *vd, const uint64_t *vc, int64_t n, int64_t *scratch,
                              int64_t *result, hipStream_t s) {
    int grid = grid_for(n, 256, 8);
    if (grid > kFoldBlocks) grid = kFoldBlocks;
    k_fold_global<<<grid, 256, 0, s>>>(kind, d, vd, vc, n, scratch);
    k_fold_global_finish<<<1, 256, 0, s>>>(kind, d, scratch, grid, result);
    return hipGetLastError();
}

// one-hot vectors {value, slot, count}: element-wise ops between fold results
__global__ void k_onehot_binary(int op, const int64_t *a, const int64_t *b, int64_t *out) {
    const bool ok = a[2] > 0 && b[2] > 0 && a[1] == b[1] && a[1] >= 0;
    out[0] = ok ? apply_bin(op, a[0], b[0]) : 0;
    out[1] = a[1];
    out[2] = ok ? 1 : 0;
}
hipError_t launch_onehot_binary(int op, const int64_t *a, const int64_t *b, int64_t *out, hipStream_t s) {
    k_onehot_binary<<<1, 1, 0, s>>>(op, a, b, out);
    return hipGetLastError();
}
__global__ void k_onehot_const(int op, const int64_t *a, int64_t k, int const_left, int64_t *out) {
    const bool ok = a[2] > 0;
    out[0] = ok ? (const_left ? apply_bin(op, k, a[0]) : apply_bin(op, a[0], k)) : 0;
    out[1] = a[1];
    out[2] = a[2];
}
hipError_t launch_onehot_const(int op, const int64_t *a, int64_t k, int const_left, int64_t *out, hipStream_t s) {
    k_onehot_const<<<1, 1, 0, s>>>(op, a, k, const_left, out);
    return hipGetLastError();
}
__global__ void k_onehot_dense(const int64_t *oh, int64_t *out, uint64_t *valid, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t slot = oh[2] > 0 ? oh[1] : -1;
    const int64_t nw = (n + 63) >> 6;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (i == slot) ? oh[0] : 0;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += stride)
        valid[w] = (slot >= 0 && (slot >> 6) == w) ? (1ull << (slot & 63)) : 0ull;
}
hipError_t launch_onehot_dense(const int64_t *oh, int64_t *out, uint64_t *valid, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    k_onehot_dense<<<grid_for(n, 256, 4), 256, 0, s>>>(oh, out, valid, n);
    return hipGetLastError();
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
    const int64_t nb = (n + compact_tile() - 1) / compact_tile();
    if (nb <= 0) return hipSuccess;
    k_compact_count<<<(int)nb, 64, 0, s>>>(valid, n, counts);
    return hipGetLastError();
}

// single-block exclusive scan (in place); total written at [nblocks]
__global__ __launch_bounds__(1024) void k_scan_counts(int64_t *c, int64_t nb) {
    __shared__ int64_t wsum[1024 / kWave];
    __shared__ int64_t carry;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t i = base + tid;
        const int64_t x = i < nb ? c[i] : 0;
        int64_t incl = x;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            int64_t y = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += y;
        }
        if (lane == kWave - 1) wsum[wave] = incl;
        __syncthreads();
        int64_t wprefix = 0;
        for (int w = 0; w < wave; w++) wprefix += wsum[w];
        const int64_t excl = carry + wprefix + incl - x;
        __syncthreads();
        if (i < nb) c[i] = excl;
        if (tid == 1023) carry = excl + x;
        __syncthreads();
    }
    if (tid == 0) c[nb] = carry;
}
hipError_t launch_compact_scan(int64_t *counts, int64_t nb, hipStream_t s) {
    k_scan_counts<<<1, 1024, 0, s>>>(counts, nb);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_compact_write(Src v, const uint64_t *valid, int64_t n, const int64_t *offsets, int64_t *out) {
    __shared__ int wcount[kCompactWords];
    __shared__ int wprefix[kCompactWords];
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
        wcount[tid] = __popcll(m);
    }
    __syncthreads();
    if (tid == 0) { int run = 0; for (int k = 0; k < kCompactWords; k++) { wprefix[k] = run; run += wcount[k]; } }
    __syncthreads();
    const int64_t base = offsets[blockIdx.x];
    for (int k = wave; k < kCompactWords; k += 256 / kWave) {
        const int64_t w = w0 + k;
        if (w >= nw) break;
        uint64_t m = valid ? valid[w] : ~0ull;
        const int64_t i = (w << 6) + lane;
        if (i < n && ((m >> lane) & 1ull)) {
            const int rank = __popcll(m & ((1ull << lane) - 1));
            out[base + wprefix[k] + rank] = ld(v, i);
        }
    }
}
hipError_t launch_compact_write(Src v, const uint64_t *valid, int64_t n, const int64_t *offsets, int64_t *out, hipStream_t s) {
    const int64_t nb = (n + compact_tile() - 1) / compact_tile();
    if (nb <= 0) return hipSuccess;
    k_compact_write<<<(int)nb, 256, 0, s>>>(v, valid, n, offsets, out);
    return hipGetLastError();
}

// Gather (/root/reference/src/Vdl.hs:438): out_i = src[pos_i]; EPS if pos_i is EPS / out of range /
// the source slot is EPS.  One ballot per wave writes the validity word.
__global__ __launch_bounds__(256) void k_gather(Src src, const uint64_t *vsrc, int64_t nsrc, Src pos, const uint64_t *vpos,
                                                int64_t n, int64_t *out, uint64_t *vout) {
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave);
    for (int64_t w = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; w < nw; w += wstride) {
        const int64_t i = (w << 6) + lane;
        bool ok = i < n && bit(vpos, i);
        int64_t p = ok ? ld(pos, i) : 0;
        ok = ok && p >= 0 && p < nsrc && bit(vsrc, p);
        if (i < n) out[i] = ok ? ld(src, p) : 0;
        const uint64_t m = __ballot(ok);
        if (lane == 0) vout[w] = m;
    }
}
hipError_t launch_gather(Src src, const uint64_t *vsrc, int64_t nsrc, Src pos, const uint64_t *vpos, int64_t n, int64_t *out,
                         uint64_t *vout, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    k_gather<<<grid_for(n, 256, 4), 256, 0, s>>>(src, vsrc, nsrc, pos, vpos, n, out, vout);
    return hipGetLastError();
}

// Scatter (/root/reference/src/Vdl.hs:441-442): out[pos_i] = src_i; positions are unique at every
// call site (/root/reference/src/Vlite.hs:1267,508), so plain stores do not race; the validity
// bitmap is set with 64-bit atomic OR.
__global__ __launch_bounds__(256) void k_scatter(Src src, const uint64_t *vsrc, Src pos, const uint64_t *vpos, int64_t n,
                                                 int64_t nout, int64_t *out, uint64_t *vout) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!bit(vsrc, i) || !bit(vpos, i)) continue;
        const int64_t p = ld(pos, i);
        if (p < 0 || p >= nout) continue;
        out[p] = ld(src, i);
        atomicOr((unsigned long long *)&vout[p >> 6], 1ull << (p & 63));
    }
}
hipError_t launch_scatter(Src src, const uint64_t *vsrc, Src pos, const uint64_t *vpos, int64_t n, int64_t nout, int64_t *out,
                          uint64_t *vout, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    k_scatter<<<grid_for(n, 256, 4), 256, 0, s>>>(src, vsrc, pos, vpos, n, nout, out, vout);
    return hipGetLastError();
}

}  // namespace vdl

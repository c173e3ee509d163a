// vdl_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4, wave64).
//
// Everything on this path is HBM-bound integer work (no MFMA anywhere: there is no dense
// contraction in a VDL program).  The rules that matter: coalesced 16-byte-per-lane column
// loads, enough bytes in flight per CU, wave-level reductions (64-lane shuffles / ballots),
// LDS only for the cross-wave step, and one pass over every byte.
#include "vdl_kernels.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace vdl {

typedef long long ll2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef char i8x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;

// hipGetLastError() is sticky per thread: a failed call made earlier by anybody in this process
// (e.g. an advisory query) would be reported by the next launch check.  Launchers therefore clear
// the slot before launching and read it right after (launch_status).
static inline hipError_t launch_status() { return hipGetLastError(); }

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
    (void)hipGetLastError();   // see launch_status()
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
    return launch_status();
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

template <bool NT, typename V>
__device__ __forceinline__ V stream_load(const char *p) {
    if (NT) return __builtin_nontemporal_load((const V *)p);
    return *(const V *)p;
}

template <int NC, int NA, int U, bool VEC, bool NT, int BS>
__global__ __launch_bounds__(BS) void k_scan(const ScanArgs A) {
    constexpr int TILE = BS * 2 * U;
    constexpr int ROWS = 2 * U;
    const int tid = threadIdx.x;
    int64_t acc[NA];
    int64_t cnt = 0;
#pragma unroll
    for (int j = 0; j < NA; j++) acc[j] = (j < A.nagg) ? r_identity(A.kind[j]) : 0;

    const int64_t ntiles = A.n / TILE;
    // tile -> block mapping: grid-stride (neighbouring blocks read neighbouring tiles) or one
    // contiguous chunk of tiles per block (A.chunked)
    int64_t tile = blockIdx.x, tile_end = ntiles, tile_step = gridDim.x;
    if (A.chunked) {
        tile = ntiles * blockIdx.x / gridDim.x;
        tile_end = ntiles * (blockIdx.x + 1) / gridDim.x;
        tile_step = 1;
    }
    for (; tile < tile_end; tile += tile_step) {
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
                        v[c][2 * u] = load_scalar(p, w, base + (int64_t)u * (BS * 2));
                        v[c][2 * u + 1] = load_scalar(p, w, base + (int64_t)u * (BS * 2) + 1);
                    }
                } else if (w == 8) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        ll2 x = stream_load<NT, ll2>(p + (base + (int64_t)u * (BS * 2)) * 8);
                        v[c][2 * u] = x.x; v[c][2 * u + 1] = x.y;
                    }
                } else if (w == 4) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        i32x2 x = stream_load<NT, i32x2>(p + (base + (int64_t)u * (BS * 2)) * 4);
                        v[c][2 * u] = x.x; v[c][2 * u + 1] = x.y;
                    }
                } else if (w == 2) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        i16x2 x = stream_load<NT, i16x2>(p + (base + (int64_t)u * (BS * 2)) * 2);
                        v[c][2 * u] = x.x; v[c][2 * u + 1] = x.y;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        i8x2 x = stream_load<NT, i8x2>(p + (base + (int64_t)u * (BS * 2)));
                        v[c][2 * u] = x.x; v[c][2 * u + 1] = x.y;
                    }
                }
            }
        }
        scan_accumulate<NC, NA, ROWS>(A, v, acc, cnt);
    }
    // tail rows (fewer than one tile) go to the last block, one row per lane
    if (blockIdx.x == gridDim.x - 1) {
        for (int64_t i = ntiles * TILE + tid; i < A.n; i += BS) {
            int64_t v1[NC][1];
#pragma unroll
            for (int c = 0; c < NC; c++)
                if (c < A.ncol) v1[c][0] = load_scalar(A.ptr[c], A.width[c], i);
            scan_accumulate<NC, NA, 1>(A, v1, acc, cnt);
        }
    }
    // block reduction: shuffles inside each wave, LDS across the waves
    __shared__ int64_t red[BS / kWave][NA + 1];
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
        for (int w = 1; w < BS / kWave; w++) x += red[w][0];
        dst[0] = x;
#pragma unroll
        for (int j = 0; j < NA; j++) {
            if (j < A.nagg) {
                int64_t y = red[0][j + 1];
#pragma unroll
                for (int w = 1; w < BS / kWave; w++) y = r_combine(A.kind[j], y, red[w][j + 1]);
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
struct ScanVariant { int nc, na, u; bool vec, nt; int bs; scan_fn fn; const char *name; };
#define VDL_SV(NC, NA, U, VEC, NT, BS) {NC, NA, U, VEC, NT, BS, k_scan<NC, NA, U, VEC, NT, BS>, "k_scan<" #NC "," #NA "," #U "," #VEC "," #NT "," #BS ">"}
// The first entry that fits (ncol, nagg, alignment) is the production kernel; the rest of the
// (4,1) family exists for tuning sweeps (VDL_SCAN_TUNE, see scan_launch_config).
const ScanVariant kScanVariants[] = {
    // production kernels (tuned on Q6 SF100, profiles/r01/tune_scan.md): non-temporal loads, one
    // wave per SIMD with 12 sub-iterations = 48 loads (42 KiB) in flight per wave
    // (several aggregates or more than 4 columns: k_mscan, vdl_mscan.hip)
    VDL_SV(4, 1, 12, true, true, 256),
    VDL_SV(4, 1, 4, false, false, 256),
    // small inputs: smaller tiles so that every CU gets work
    VDL_SV(4, 1, 4, true, true, 256),
    // tuning family for sweeps (VDL_SCAN_TUNE)
    VDL_SV(4, 1, 2, true, false, 256), VDL_SV(4, 1, 4, true, false, 256), VDL_SV(4, 1, 8, true, false, 256),
    VDL_SV(4, 1, 2, true, true, 256),  VDL_SV(4, 1, 8, true, true, 256),
    VDL_SV(4, 1, 2, true, false, 512), VDL_SV(4, 1, 4, true, false, 512), VDL_SV(4, 1, 8, true, false, 512),
    VDL_SV(4, 1, 2, true, true, 512),  VDL_SV(4, 1, 4, true, true, 512),  VDL_SV(4, 1, 8, true, true, 512),
    VDL_SV(4, 1, 10, true, true, 256), VDL_SV(4, 1, 14, true, true, 256), VDL_SV(4, 1, 12, true, false, 256), VDL_SV(4, 1, 6, true, true, 512),
    VDL_SV(4, 1, 2, true, false, 1024), VDL_SV(4, 1, 2, true, true, 1024),
};
#undef VDL_SV
constexpr int kNumScanVariants = sizeof(kScanVariants) / sizeof(kScanVariants[0]);

int tune_value(const char *spec, const char *key, int dflt) {
    if (!spec) return dflt;
    const char *p = strstr(spec, key);
    if (!p) return dflt;
    p += strlen(key);
    if (*p != '=') return dflt;
    return atoi(p + 1);
}
}  // namespace

ScanLaunch scan_launch_config(ScanArgs &a, int num_cus) {
    // vector loads need every column base aligned to its two-row access
    bool vec = true;
    for (int c = 0; c < a.ncol; c++)
        if (((uintptr_t)a.ptr[c]) % (uintptr_t)(2 * a.width[c]) != 0) vec = false;
    // Tuning knob for sweeps (tools/tune_scan.py): VDL_SCAN_TUNE="u=8,nt=1,bs=512,gridmul=2,percu=4,chunk=1"
    const char *tune = getenv("VDL_SCAN_TUNE");
    const int want_u = tune_value(tune, "u", -1), want_nt = tune_value(tune, "nt", -1), want_bs = tune_value(tune, "bs", -1);
    ScanLaunch cfg;
    cfg.variant = -1;
    for (int i = 0; i < kNumScanVariants; i++) {
        const ScanVariant &v = kScanVariants[i];
        if (v.vec != vec || a.ncol > v.nc || a.nagg > v.na) continue;
        if (tune && vec && ((want_u >= 0 && v.u != want_u) || (want_nt >= 0 && (int)v.nt != want_nt) || (want_bs >= 0 && v.bs != want_bs))) continue;
        // a big-tile kernel needs a few tiles per CU to balance; smaller inputs take the next fitting variant
        if (!tune && vec && v.u > 4 && a.n / ((int64_t)v.bs * 2 * v.u) < 4 * (int64_t)num_cus) continue;
        cfg.variant = i;
        break;
    }
    if (cfg.variant < 0 && tune) {     // no tuning variant of this shape: fall back to production
        for (int i = 0; i < kNumScanVariants; i++) {
            const ScanVariant &v = kScanVariants[i];
            if (v.vec == vec && a.ncol <= v.nc && a.nagg <= v.na) { cfg.variant = i; break; }
        }
    }
    if (cfg.variant < 0) return cfg;
    const ScanVariant &v = kScanVariants[cfg.variant];
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v.fn, v.bs, 0) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();      // the query is advisory: do not leave its error for the next launch check
        per_cu = 4;
    }
    if (per_cu > 8) per_cu = 8;
    per_cu = tune_value(tune, "percu", per_cu);
    const int64_t tile = (int64_t)v.bs * 2 * v.u;
    const int64_t ntiles = a.n / tile;
    int64_t grid = (int64_t)num_cus * per_cu * tune_value(tune, "gridmul", 1);
    if (grid > ntiles) grid = ntiles;
    if (grid < 1) grid = 1;
    cfg.grid = (int)grid;
    cfg.block = v.bs;
    a.chunked = tune_value(tune, "chunk", 0);
    return cfg;
}

const char *scan_kernel_name(const ScanLaunch &cfg) {
    return (cfg.variant >= 0 && cfg.variant < kNumScanVariants) ? kScanVariants[cfg.variant].name : "none";
}

hipError_t launch_scan(const ScanArgs &a, const ScanLaunch &cfg, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    if (cfg.variant < 0 || cfg.variant >= kNumScanVariants) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kScanVariants[cfg.variant].fn, dim3(cfg.grid), dim3(cfg.block), 0, s, a);
    return launch_status();
}

hipError_t launch_scan_finish(const int64_t *block_partials, int nblocks, int, const int *, const ScanArgs &a, int64_t *words,
                              hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    k_scan_finish<<<1, 256, 0, s>>>(block_partials, nblocks, a, words);
    return launch_status();
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

// Element loads with the representation fixed at compile time, and a dispatcher that runs a body once with the
// wave-uniform kind turned into a constant.  Written with ld()/bit() and `&&`, a gather is a chain of exec-masked
// regions, each load waited for before the next test; with unconditional loads (masked lanes read slot 0) and bitwise
// tests the loads of several bitmap words are in flight together.
template <int K> __device__ __forceinline__ int64_t ldk(const Src &s, int64_t i) {
    if (K == SRC_I64) return ((const int64_t *)s.p)[i];
    if (K == SRC_I32) return ((const int32_t *)s.p)[i];
    if (K == SRC_I16) return ((const int16_t *)s.p)[i];
    if (K == SRC_I8) return ((const int8_t *)s.p)[i];
    return (int64_t)((uint64_t)s.from + (uint64_t)i * (uint64_t)s.step);
}
template <class F> __device__ __forceinline__ void by_kind(int kind, F f) {
    switch (kind) {
    case SRC_I64: f(std::integral_constant<int, SRC_I64>{}); break;
    case SRC_I32: f(std::integral_constant<int, SRC_I32>{}); break;
    case SRC_I16: f(std::integral_constant<int, SRC_I16>{}); break;
    case SRC_I8: f(std::integral_constant<int, SRC_I8>{}); break;
    default: f(std::integral_constant<int, SRC_RANGE>{}); break;
    }
}
constexpr int kGatherUnroll = 4;      // bitmap words (64 positions each) a wave has in flight

// positions of U consecutive words -> clamped source slots pc[] and lane flags ok[] (position present, in range,
// source slot holds a value); `extra` = a second validity bitmap over the positions (may be null)
template <int KP, bool VS>
__device__ __forceinline__ void gather_slots(const Src &pos, const uint64_t *vpos, const uint64_t *extra, const uint64_t *vsrc, int64_t nsrc,
                                             int64_t n, int64_t nw, int64_t w0, int lane, int64_t (&pc)[kGatherUnroll], bool (&ok)[kGatherUnroll]) {
    constexpr int U = kGatherUnroll;
    int64_t p[U], w[U];
    uint64_t a[U];
    bool in[U];
#pragma unroll
    for (int u = 0; u < U; u++) {                                  // the position loads go out first ...
        w[u] = w0 + u < nw ? w0 + u : nw - 1;                      // wave-uniform; spare words repeat the last one and are not stored
        a[u] = w0 + u < nw ? ~0ull : 0ull;
        const int64_t i = (w[u] << 6) + lane;
        in[u] = i < n;
        p[u] = ldk<KP>(pos, in[u] ? i : 0);
    }
    if (vpos) {                                                    // ... then the bitmap words, U loads under one branch
        uint64_t t[U];
#pragma unroll
        for (int u = 0; u < U; u++) t[u] = vpos[w[u]];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] &= t[u];
    }
    if (extra) {
        uint64_t t[U];
#pragma unroll
        for (int u = 0; u < U; u++) t[u] = extra[w[u]];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] &= t[u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) ok[u] = in[u] & (((a[u] >> lane) & 1ull) != 0);
#pragma unroll
    for (int u = 0; u < U; u++) {
        ok[u] = ok[u] & (p[u] >= 0) & (p[u] < nsrc);
        pc[u] = ok[u] ? p[u] : 0;
    }
    if (VS) {
        uint64_t vw[U];
#pragma unroll
        for (int u = 0; u < U; u++) vw[u] = vsrc[pc[u] >> 6];
#pragma unroll
        for (int u = 0; u < U; u++) ok[u] = ok[u] & (((vw[u] >> (pc[u] & 63)) & 1ull) != 0);
    }
}
__device__ __forceinline__ int64_t wave_index() {               // in an SGPR: the bitmap words of a wave are scalar loads
    return (int64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave));
}

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
constexpr int kExprRows = 4;
#define VDL_EX_PUSH(K) case K: _Pragma("unroll") for (int r = 0; r < kExprRows; r++) st[K][r] = row[r] < n ? ld(lf, row[r]) : 0; break;
// a (x) b for the lane's rows; the operator switch is wave-uniform and sits outside the row loop
#define VDL_EX_OP(OP) case OP: _Pragma("unroll") for (int r = 0; r < kExprRows; r++) a[r] = apply_bin(OP, a[r], b[r]); break;
__device__ __forceinline__ void expr_rows(int op, int64_t (&a)[kExprRows], const int64_t (&b)[kExprRows]) {
    switch (op) {
        VDL_EX_OP(B_LAND) VDL_EX_OP(B_LOR) VDL_EX_OP(B_BAND) VDL_EX_OP(B_BOR) VDL_EX_OP(B_SHIFT) VDL_EX_OP(B_EQ)
        VDL_EX_OP(B_ADD) VDL_EX_OP(B_SUB) VDL_EX_OP(B_GT) VDL_EX_OP(B_MUL)
        case X_GE: _Pragma("unroll") for (int r = 0; r < kExprRows; r++) a[r] = a[r] >= b[r]; break;
        case X_NE: _Pragma("unroll") for (int r = 0; r < kExprRows; r++) a[r] = a[r] != b[r]; break;
        // Divide / Modulo are kept out of fused trees (vdl_engine.cpp expr_binary): the 64-bit division routine inlined
        // at every stack height tripled the size of the kernel
    }
}
#undef VDL_EX_OP
#define VDL_EX_BIN(K) case K: expr_rows(op, st[K - 2], st[K - 1]); break;
__global__ __launch_bounds__(256) void k_expr(const ExprProg P, int64_t *__restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; base < n; base += kExprRows * stride) {
        int64_t row[kExprRows];
#pragma unroll
        for (int r = 0; r < kExprRows; r++) row[r] = base + r * stride;
        int64_t st[kExprDepth][kExprRows];
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
        for (int r = 0; r < kExprRows; r++) if (row[r] < n) out[row[r]] = st[0][r];
    }
}
#undef VDL_EX_PUSH
#undef VDL_EX_BIN
hipError_t launch_expr(const ExprProg &prog, int64_t *out, int64_t n, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    k_expr<<<grid_for(n, 256, kExprRows), 256, 0, s>>>(prog, out, n);
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
hipError_t launch_filter_columns(const FilterArgs &a0, uint64_t *out, int64_t n, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    FilterArgs a = a0;
    int ni = 1;
    for (int c = 0; c < a.ncol; c++) ni = a.nint[c] > ni ? a.nint[c] : ni;
    ni = ni <= 1 ? 1 : ni <= 2 ? 2 : kMaxFilterIvs;
    for (int c = 0; c < a.ncol; c++)
        for (int k = a.nint[c]; k < kMaxFilterIvs; k++) { a.lo[c][k] = 1; a.hi[c][k] = 0; }       // empty
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
    (void)hipGetLastError();   // see launch_status()
    if (n <= 0) return hipSuccess;
    k_select_bitmap<<<grid_for(n, 256, 4), 256, 0, s>>>(d, vd, vc, out, n);
    return launch_status();
}

// Global (single-run) fold (/root/reference/src/Vlite.hs:337-356 with an all-equal control
// vector, Vlite.hs:636-639): two launches, per-block partials then one block.
constexpr int kFoldBlocks = 2048;
int fold_scratch_blocks() { return kFoldBlocks; }

__global__ __launch_bounds__(256) void k_fold_global(int kind, Src d, const uint64_t *vd, const uint64_t *vc, int64_t n,
                                                     int64_t *scratch) {
    constexpr int U = kGatherUnroll;
    const int rk = kind == 1 ? R_MIN : kind == 2 ? R_MAX : R_SUM;
    const int ak = kind == 4 ? R_MIN : rk;
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave) * U;
    // per lane: the fold of its data; per wave (kept uniform, lane 0 reports them): first control slot, number of data
    // slots, and for count / choose the result itself -- all three come from the bitmap words, not from the rows
    int64_t acc = r_identity(rk);
    int64_t first = INT64_MAX, cnt = 0, chosen = INT64_MAX;
    by_kind(d.kind, [&](auto kd) {
        for (int64_t w0 = wave_index() * U; w0 < nw; w0 += wstride) {
            int64_t x[U];
            uint64_t mc[U], md[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t w = w0 + u < nw ? w0 + u : nw - 1;
                const int64_t i = (w << 6) + lane;
                x[u] = kind < 3 ? ldk<decltype(kd)::value>(d, i < n ? i : 0) : 0;     // wave-uniform test; masked lanes read slot 0
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
                if ((md[u] >> lane) & 1ull) acc = r_combine(rk, acc, x[u]);
            }
        }
    });
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
        result[1] = first == INT64_MAX ? -1 : first;
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
__global__ void k_onehot_dense(const int64_t *oh, int64_t *out, uint64_t *valid, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t slot = oh[2] > 0 ? oh[1] : -1;
    const int64_t nw = (n + 63) >> 6;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (i == slot) ? oh[0] : 0;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += stride)
        valid[w] = (slot >= 0 && (slot >> 6) == w) ? (1ull << (slot & 63)) : 0ull;
}
hipError_t launch_onehot_dense(const int64_t *oh, int64_t *out, uint64_t *valid, int64_t n, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    if (n <= 0) return hipSuccess;
    k_onehot_dense<<<grid_for(n, 256, 4), 256, 0, s>>>(oh, out, valid, n);
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
__global__ __launch_bounds__(1024) void k_scan_counts(int64_t *c, int64_t nb) {
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
    if (tid == 0) c[nb] = carry;
}
hipError_t launch_compact_scan(int64_t *counts, int64_t nb, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    k_scan_counts<<<1, 1024, 0, s>>>(counts, nb);
    return launch_status();
}

__global__ __launch_bounds__(256) void k_compact_write(Src v, const uint64_t *valid, int64_t n, const int64_t *offsets, int64_t *out) {
    __shared__ int wcount[kCompactWords];
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
        wcount[tid] = __popcll(m);
    }
    __syncthreads();
    if (tid == 0) { int run = 0; for (int k = 0; k < kCompactWords; k++) { wprefix[k] = run; run += wcount[k]; } }
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
                }
            }
        }
    });
}
hipError_t launch_compact_write(Src v, const uint64_t *valid, int64_t n, const int64_t *offsets, int64_t *out, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    const int64_t nb = (n + compact_tile() - 1) / compact_tile();
    if (nb <= 0) return hipSuccess;
    k_compact_write<<<(int)nb, 256, 0, s>>>(v, valid, n, offsets, out);
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
hipError_t launch_select_gather(Src src, const uint64_t *vsrc, int64_t nsrc, Src pos, const uint64_t *vpos, const uint64_t *vc, uint64_t *out,
                                int64_t n, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    k_select_gather<<<grid_for(n, 256, 4), 256, 0, s>>>(src, vsrc, nsrc, pos, vpos, vc, out, n);
    return launch_status();
}

__global__ __launch_bounds__(256) void k_set_bits(const int64_t *idx, int64_t m, uint64_t *bitmap) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) {
        const int64_t i = idx[k];
        atomicOr((unsigned long long *)&bitmap[i >> 6], 1ull << (i & 63));
    }
}
hipError_t launch_set_bits(const int64_t *idx, int64_t m, uint64_t *bitmap, hipStream_t s) {
    (void)hipGetLastError();
    if (m <= 0) return hipSuccess;
    k_set_bits<<<grid_for(m, 256, 4), 256, 0, s>>>(idx, m, bitmap);
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
// device-wide exclusive prefix sum over int64 (in place): block sums -> one-block scan of the
// sums -> per-block scan with carry.  Tile = 1024 threads x 4 consecutive items.
// ------------------------------------------------------------------------------------------
constexpr int kPsBlock = 1024, kPsItems = 4, kPsTile = kPsBlock * kPsItems;
int64_t prefix_sum_blocks(int64_t n) { return (n + kPsTile - 1) / kPsTile; }

__global__ __launch_bounds__(kPsBlock) void k_ps_block_sums(const int64_t *x, int64_t n, int64_t *sums) {
    __shared__ int64_t red[kPsBlock / kWave];
    const int64_t base = (int64_t)blockIdx.x * kPsTile + (int64_t)threadIdx.x * kPsItems;
    int64_t t = 0;
#pragma unroll
    for (int k = 0; k < kPsItems; k++) if (base + k < n) t += x[base + k];
    t = wave_reduce(t, R_SUM);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = t;
    __syncthreads();
    if (threadIdx.x == 0) { int64_t a = 0; for (int w = 0; w < kPsBlock / kWave; w++) a += red[w]; sums[blockIdx.x] = a; }
}

__global__ __launch_bounds__(kPsBlock) void k_ps_apply(int64_t *x, int64_t n, const int64_t *block_excl) {
    __shared__ int64_t wsum[kPsBlock / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int64_t base = (int64_t)blockIdx.x * kPsTile + (int64_t)tid * kPsItems;
    int64_t v[kPsItems], t = 0;
#pragma unroll
    for (int k = 0; k < kPsItems; k++) { v[k] = (base + k < n) ? x[base + k] : 0; t += v[k]; }
    int64_t incl = t;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) { int64_t y = __shfl_up(incl, off, kWave); if (lane >= off) incl += y; }
    if (lane == kWave - 1) wsum[wave] = incl;
    __syncthreads();
    int64_t run = block_excl[blockIdx.x] + incl - t;
    for (int w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
    for (int k = 0; k < kPsItems; k++) { if (base + k < n) x[base + k] = run; run += v[k]; }
}

// sums: prefix_sum_blocks(n) + 1 int64 of scratch; the grand total is left in sums[nblocks]
hipError_t launch_prefix_sum(int64_t *x, int64_t n, int64_t *sums, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    const int64_t nb = prefix_sum_blocks(n);
    if (nb <= 0) return hipSuccess;
    k_ps_block_sums<<<(int)nb, kPsBlock, 0, s>>>(x, n, sums);
    k_scan_counts<<<1, 1024, 0, s>>>(sums, nb);
    k_ps_apply<<<(int)nb, kPsBlock, 0, s>>>(x, n, sums);
    return launch_status();
}

// ------------------------------------------------------------------------------------------
// Partition (/root/reference/src/Vdl.hs:130,266-269; Vlite.hs:358-366,508,1082-1098): positions
// that stably group `data` by pivot bucket.  Pivots are the emitted RangeC min cnt 1, so
// bucket = clamp(data - min, 0, cnt).  Implemented as an LSD radix sort of (bucket, slot) over the
// non-EPS slots, 8 bits per pass; each pass = tile histogram (LDS atomics) -> device-wide prefix
// sum in digit-major order -> stable scatter (see k_part_scatter).  The last pass writes out[slot] = rank instead of the sorted pair.
// Dense group-by domains (Q1: 32 buckets) need one pass, Q3's 2^38 domain five.
// ------------------------------------------------------------------------------------------
constexpr int kPartBlock = 256, kPartSteps = 16, kPartTile = kPartBlock * kPartSteps, kRadix = 256;
int64_t partition_tiles(int64_t n) { return (n + kPartTile - 1) / kPartTile; }

struct PartIn {
    Src data;                    // first pass: raw data column
    const uint64_t *valid;       // first pass: validity of data
    int64_t pmin, pcount;        // pivots = RangeC pmin pcount 1
    const uint64_t *keys;        // later passes: bucket values of the previous pass
    const int64_t *slots;        // later passes: originating slot
    const int64_t *n_dev;        // later passes: number of elements (device scalar)
    int64_t n;                   // first pass: number of slots; later: upper bound for the grid
    int shift;
};

// A wave's share of a tile: kPartSteps x 64 consecutive slots starting at a multiple of 64, fetched with every load
// issued before the first use (the validity word of a step is the same for all lanes).
template <bool FIRST>
__device__ __forceinline__ void part_fetch_share(const PartIn &in, int64_t n, int64_t share0 /* multiple of 64 */, int lane,
                                                 uint64_t (&keys)[kPartSteps], int64_t (&slots)[kPartSteps], bool (&oks)[kPartSteps]) {
    if (FIRST) {
        by_kind(in.data.kind, [&](auto kd) {
#pragma unroll
            for (int st = 0; st < kPartSteps; st++) {
                const int64_t i = share0 + st * kWave + lane;
                oks[st] = i < n;
                slots[st] = i;
                keys[st] = (uint64_t)ldk<decltype(kd)::value>(in.data, oks[st] ? i : 0);
            }
        });
        if (in.valid) {
            const int64_t nw = (n + 63) >> 6;
            uint64_t t[kPartSteps];
#pragma unroll
            for (int st = 0; st < kPartSteps; st++) { const int64_t w = (share0 >> 6) + st; t[st] = in.valid[w < nw ? w : nw - 1]; }
#pragma unroll
            for (int st = 0; st < kPartSteps; st++) oks[st] = oks[st] & (((t[st] >> lane) & 1ull) != 0);
        }
#pragma unroll
        for (int st = 0; st < kPartSteps; st++) {                // bucket = clamp(data - min, 0, cnt)
            const int64_t x = (int64_t)keys[st];
            int64_t b = 0;
            if (x > in.pmin) { b = (int64_t)((uint64_t)x - (uint64_t)in.pmin); if (b < 0 || b > in.pcount) b = in.pcount; }
            keys[st] = (uint64_t)b;
        }
    } else {
#pragma unroll
        for (int st = 0; st < kPartSteps; st++) {
            const int64_t i = share0 + st * kWave + lane;
            oks[st] = i < n;
            const int64_t ii = oks[st] ? i : 0;                  // n > 0 here
            keys[st] = in.keys[ii]; slots[st] = in.slots[ii];
        }
    }
}

template <bool FIRST>
__global__ __launch_bounds__(kPartBlock) void k_part_hist(PartIn in, int64_t ntiles, int64_t *hist /*[256][ntiles]*/) {
    __shared__ unsigned int h[kRadix];
    const int64_t n = FIRST ? in.n : *in.n_dev;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    h[tid] = 0;
    __syncthreads();
    if ((int64_t)blockIdx.x * kPartTile < n) {                   // (n = 0 leaves the key buffers unwritten)
        uint64_t keys[kPartSteps];
        int64_t slots[kPartSteps];
        bool oks[kPartSteps];
        part_fetch_share<FIRST>(in, n, (int64_t)blockIdx.x * kPartTile + (int64_t)wave * (kPartSteps * kWave), lane, keys, slots, oks);
#pragma unroll
        for (int st = 0; st < kPartSteps; st++)
            if (oks[st]) atomicAdd(&h[(keys[st] >> in.shift) & (kRadix - 1)], 1u);
    }
    __syncthreads();
    hist[(int64_t)tid * ntiles + blockIdx.x] = h[tid];
}

// Each wave owns a contiguous quarter of the tile (16 steps of 64 slots), so the stable order inside a tile is wave,
// step, lane.  A wave ranks its slots on its own (peer masks from 8 ballots, its running digit counts in its row of
// whist: LDS operations of one wave execute in order); one barrier later the rows are turned into per-wave offsets
// and every slot knows its destination.  (A version that kept the block in step order needed three barriers per
// step, 48 per tile, and was twice as slow.)
template <bool FIRST, bool LAST>
__global__ __launch_bounds__(kPartBlock) void k_part_scatter(PartIn in, int64_t ntiles, const int64_t *offsets,
                                                              uint64_t *keys_out, int64_t *slots_out, int64_t *pos_out) {
    constexpr int NW = kPartBlock / kWave;
    __shared__ int64_t woff[NW][kRadix];
    __shared__ unsigned int whist[NW][kRadix];
    const int64_t n = FIRST ? in.n : *in.n_dev;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
#pragma unroll
    for (int w = 0; w < NW; w++) whist[w][tid] = 0;
    __syncthreads();
    uint64_t keys[kPartSteps];
    int64_t slots[kPartSteps];
    bool oks[kPartSteps];
#pragma unroll
    for (int st = 0; st < kPartSteps; st++) { keys[st] = 0; slots[st] = 0; oks[st] = false; }
    if ((int64_t)blockIdx.x * kPartTile < n)                    // the whole share is fetched up front
        part_fetch_share<FIRST>(in, n, (int64_t)blockIdx.x * kPartTile + (int64_t)wave * (kPartSteps * kWave), lane, keys, slots, oks);
    unsigned int local[kPartSteps];                             // rank among this wave's slots with the same digit
    volatile unsigned int *mine = whist[wave];
#pragma unroll
    for (int st = 0; st < kPartSteps; st++) {
        const unsigned d = (unsigned)((keys[st] >> in.shift) & (kRadix - 1));
        uint64_t peers = __ballot(oks[st]);                     // lanes of this wave holding the same digit (and a value)
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const unsigned rank = (unsigned)__popcll(peers & ((1ull << lane) - 1));
        const unsigned pre = oks[st] ? mine[d] : 0u;
        if (oks[st] && rank == 0) mine[d] = pre + (unsigned)__popcll(peers);     // one leader per digit
        local[st] = pre + rank;
    }
    __syncthreads();
    if (LAST) {                                                 // ranks go to out[slot]: scattered whatever the order
        {
            int64_t run = offsets[(int64_t)tid * ntiles + blockIdx.x];          // where this tile's slots of digit `tid` start
#pragma unroll
            for (int w = 0; w < NW; w++) { woff[w][tid] = run; run += whist[w][tid]; }
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < kPartSteps; st++) {
            if (oks[st]) {
                const unsigned d = (unsigned)((keys[st] >> in.shift) & (kRadix - 1));
                pos_out[slots[st]] = woff[wave][d] + local[st];
            }
        }
        return;
    }
    // The tile is put into digit order in LDS first (keys, then slots through the same buffer): a digit's slots of
    // one tile are neighbours at the destination, so consecutive lanes then store consecutive words instead of 64
    // scattered ones.
    __shared__ uint64_t stage[kPartTile];
    __shared__ unsigned int wtot[NW];
    __shared__ int64_t gdelta[kRadix];                          // destination of sorted index idx with digit d = gdelta[d] + idx
    unsigned total;
    {
        unsigned cnt = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) cnt += whist[w][tid];
        unsigned incl = cnt;                                    // exclusive scan of the tile's digit counts over the block
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) { const unsigned y = __shfl_up(incl, off, kWave); if (lane >= off) incl += y; }
        if (lane == kWave - 1) wtot[wave] = incl;
        __syncthreads();
        unsigned pre = 0;
        for (int w = 0; w < wave; w++) pre += wtot[w];
        total = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) total += wtot[w];
        const unsigned tstart = pre + incl - cnt;               // where digit `tid` starts inside the sorted tile
        unsigned run = tstart;
#pragma unroll
        for (int w = 0; w < NW; w++) { woff[w][tid] = run; run += whist[w][tid]; }                            // tile-local
        gdelta[tid] = offsets[(int64_t)tid * ntiles + blockIdx.x] - (int64_t)tstart;
        __syncthreads();
    }
    unsigned lpos[kPartSteps];
#pragma unroll
    for (int st = 0; st < kPartSteps; st++) {
        const unsigned d = (unsigned)((keys[st] >> in.shift) & (kRadix - 1));
        lpos[st] = (unsigned)woff[wave][d] + local[st];
        if (oks[st]) stage[lpos[st]] = keys[st];
    }
    __syncthreads();
    int64_t dest[kPartSteps];
#pragma unroll
    for (int k = 0; k < kPartSteps; k++) {
        const unsigned idx = (unsigned)k * kPartBlock + tid;
        dest[k] = -1;
        if (idx < total) {
            const uint64_t key = stage[idx];
            dest[k] = gdelta[(key >> in.shift) & (kRadix - 1)] + idx;
            keys_out[dest[k]] = key;
        }
    }
    __syncthreads();
#pragma unroll
    for (int st = 0; st < kPartSteps; st++)
        if (oks[st]) stage[lpos[st]] = (uint64_t)slots[st];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPartSteps; k++)
        if (dest[k] >= 0) slots_out[dest[k]] = (int64_t)stage[(unsigned)k * kPartBlock + tid];
}

// scratch layout is owned by the caller (vdl_engine.cpp); see launch_partition's arguments.
hipError_t launch_partition(Src data, const uint64_t *valid, int64_t n, int64_t pmin, int64_t pcount,
                            int64_t *hist /* 256*ntiles + 1 */, int64_t *scan_scratch /* prefix_sum_blocks(256*ntiles)+1 */,
                            uint64_t *keys_a, int64_t *slots_a, uint64_t *keys_b, int64_t *slots_b /* n each, or null if one pass */,
                            int64_t *n_valid_dev /* 1 word */, int64_t *pos_out, hipStream_t s) {
    (void)hipGetLastError();   // see launch_status()
    if (n <= 0) return hipSuccess;
    int bits = 0;
    while (bits < 63 && ((uint64_t)pcount >> bits) != 0) bits++;       // buckets 0..pcount
    const int passes = bits <= 8 ? 1 : (bits + 7) / 8;
    const int64_t ntiles = partition_tiles(n);
    const int64_t hn = (int64_t)kRadix * ntiles;
    PartIn in{};
    in.data = data; in.valid = valid; in.pmin = pmin; in.pcount = pcount; in.n = n; in.n_dev = n_valid_dev;
    uint64_t *kin = nullptr, *kout = keys_a; int64_t *sin = nullptr, *sout = slots_a;
    for (int p = 0; p < passes; p++) {
        in.shift = 8 * p; in.keys = kin; in.slots = sin;
        const bool first = p == 0, last = p == passes - 1;
        if (first) k_part_hist<true><<<(int)ntiles, kPartBlock, 0, s>>>(in, ntiles, hist);
        else k_part_hist<false><<<(int)ntiles, kPartBlock, 0, s>>>(in, ntiles, hist);
        hipError_t e = launch_prefix_sum(hist, hn, scan_scratch, s);
        if (e != hipSuccess) return e;
        if (first) {   // number of non-EPS slots = grand total of the first histogram
            e = hipMemcpyAsync(n_valid_dev, scan_scratch + prefix_sum_blocks(hn), sizeof(int64_t), hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return e;
        }
        if (first && last) k_part_scatter<true, true><<<(int)ntiles, kPartBlock, 0, s>>>(in, ntiles, hist, nullptr, nullptr, pos_out);
        else if (first) k_part_scatter<true, false><<<(int)ntiles, kPartBlock, 0, s>>>(in, ntiles, hist, kout, sout, nullptr);
        else if (last) k_part_scatter<false, true><<<(int)ntiles, kPartBlock, 0, s>>>(in, ntiles, hist, nullptr, nullptr, pos_out);
        else k_part_scatter<false, false><<<(int)ntiles, kPartBlock, 0, s>>>(in, ntiles, hist, kout, sout, nullptr);
        kin = kout; sin = sout;
        kout = (kout == keys_a) ? keys_b : keys_a;
        sout = (sout == slots_a) ? slots_b : slots_a;
    }
    return launch_status();
}
int partition_passes(int64_t pcount) {
    int bits = 0;
    while (bits < 63 && ((uint64_t)pcount >> bits) != 0) bits++;
    return bits <= 8 ? 1 : (bits + 7) / 8;
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
    const int rk = (kind == 1 || kind == 4) ? R_MIN : kind == 2 ? R_MAX : R_SUM;
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
    const int64_t g = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    const int64_t per = (nw + nwaves - 1) / nwaves;
    const int64_t w_end = (g + 1) * per < nw ? (g + 1) * per : nw;
    int64_t carry_h = -1, carry_x = r_identity(rk);           // wave-uniform: the run still open after the previous word
    for (int64_t w = g * per; w < w_end; w++) {
        const int64_t i = (w << 6) + lane;
        const uint64_t hm = heads[w] & (lane == 63 ? ~0ull : ((2ull << lane) - 1));   // heads at or before this lane
        int64_t h = hm ? (w << 6) + 63 - __clzll((long long)hm) : (w > 0 ? wordhd[w - 1] : -1);
        const bool ok = i < n && bit(vc, i) && bit(vd, i) && h >= 0;
        int64_t x = r_identity(rk);
        if (ok) x = kind == 3 ? 1 : kind == 4 ? i : ld(d, i);
        // lane segments = maximal stretches of consecutive active lanes with one head; an EPS slot
        // inside a run splits it into several segments, each adds its part to the same out[h]
        const int64_t hp = __shfl_up(h, 1, kWave);
        const bool okp = __shfl_up((int)ok, 1, kWave) != 0;
        const bool starts = lane == 0 || !ok || !okp || hp != h;
        const uint64_t sm = __ballot(starts) & (lane == 63 ? ~0ull : ((2ull << lane) - 1));
        const int seg0 = 63 - __clzll((long long)sm);   // first lane of my segment (bit 0 is always set)
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const int64_t y = __shfl_up(x, off, kWave);
            if (lane - off >= seg0) x = r_combine(rk, x, y);
        }
        const int64_t hn = __shfl_down(h, 1, kWave);
        const bool okn = __shfl_down((int)ok, 1, kWave) != 0;
        const bool tail = ok && (lane == kWave - 1 || !okn || hn != h);
        // runs that begin and end inside this word with every slot taking part are one segment nobody else adds to:
        // a plain store and one validity update per word (a sparse GROUP BY has ~30 such runs per word, and their
        // atomics on out[] and on the same word of vout[] were most of the kernel)
        const uint64_t hw = heads[w], okm = __ballot(ok);
        bool whole = false;
        if ((hw >> lane) & 1) {
            const uint64_t later = lane == kWave - 1 ? 0 : hw >> (lane + 1);
            if (later) {
                const int q = lane + __ffsll((long long)later);           // lane of the next head (<= 63)
                const uint64_t range = (1ull << q) - (1ull << lane);
                whole = (okm & range) == range;
            }
        }
        const uint64_t wholem = __ballot(whole);
        const bool mine = tail && ((wholem >> seg0) & 1);                  // my segment is such a run
        // the run carried over from the previous word: continue it in this word's first segment, or write it out
        if (carry_h >= 0) {
            const bool ok0 = __shfl((int)ok, 0, kWave) != 0;
            const int64_t h0 = __shfl(h, 0, kWave);
            if (ok0 && h0 == carry_h) {
                if (tail && seg0 == 0) x = r_combine(rk, x, carry_x);
            } else if (lane == 0) {
                atomic_combine(rk, &out[carry_h], carry_x);
                atomicOr((unsigned long long *)&vout[carry_h >> 6], 1ull << (carry_h & 63));
            }
            carry_h = -1;
        }
        const bool ok63 = __shfl((int)ok, kWave - 1, kWave) != 0;
        if (ok63) { carry_h = __shfl(h, kWave - 1, kWave); carry_x = __shfl(x, kWave - 1, kWave); }   // lane 63 is that segment's tail
        if (mine) {
            out[h] = x;
        } else if (tail && lane != kWave - 1) {
            atomic_combine(rk, &out[h], x);
            atomicOr((unsigned long long *)&vout[h >> 6], 1ull << (h & 63));
        }
        if (lane == 0 && wholem) atomicOr((unsigned long long *)&vout[w], wholem);   // runs from other words may set bits here too
    }
    if (carry_h >= 0 && lane == 0) {
        atomic_combine(rk, &out[carry_h], carry_x);
        atomicOr((unsigned long long *)&vout[carry_h >> 6], 1ull << (carry_h & 63));
    }
}

__global__ void k_seg_fill(int64_t *out, int64_t v, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = v;
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
    k_seg_heads<<<grid_for(n, 256, 4), 256, 0, s>>>(ctl, vc, n, heads);
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
    k_seg_fill<<<grid_for(n, 256, 4), 256, 0, s>>>(out, rk == R_SUM ? 0 : rk == R_MIN ? INT64_MAX : INT64_MIN, n);
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

// ------------------------------------------------------------------------------------------
// Row exchange for sharded Partition (Partition / join redistribution over
// xGMI).  Each rank sends every row of the partition key and of the vectors scattered by it to the
// rank that owns the row's key range; afterwards Partition / Scatter / Fold run locally on the
// received rows and the outputs of the ranks concatenate in rank order (keys ascend across ranks).
//   k_ex_dest   : destination rank of each row = (key - pmin) * world / pcount, EPS for rows that do
//                 not take part; per-destination row counts by 64-bit atomics (world <= 256)
//   (stable order inside each destination: launch_partition over the destination vector)
//   k_ex_pack   : scatter a column into send order; k_ex_mask: validity word of the source vectors
//   k_ex_unmask : received validity words -> one bitmap per source vector
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ex_dest(Src key, const uint64_t *vkey, int64_t n, int64_t pmin, int64_t pcount, int world,
                                                 int64_t *dest, uint64_t *vdest, int64_t *counts, int64_t *oob) {
    __shared__ unsigned long long cnt[kMaxExWorld + 1];       // per-block row counts per destination (+ out-of-range keys)
    for (int i = threadIdx.x; i <= kMaxExWorld; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave);
    for (int64_t w = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; w < nw; w += wstride) {
        const int64_t i = (w << 6) + lane;
        bool ok = i < n && bit(vkey, i);
        bool out_of_range = false;
        int64_t d = 0;
        if (ok) {
            const int64_t b = (int64_t)((uint64_t)ld(key, i) - (uint64_t)pmin);
            if (b < 0 || b >= pcount) { out_of_range = true; ok = false; }
            else d = (int64_t)(((unsigned __int128)(uint64_t)b * (uint64_t)world) / (uint64_t)pcount);
        }
        if (i < n) dest[i] = d;
        const uint64_t m = __ballot(ok);
        if (lane == 0) vdest[w] = m;
        const uint64_t bad = __ballot(out_of_range);
        if (bad && lane == 0) atomicAdd(&cnt[kMaxExWorld], (unsigned long long)__popcll(bad));
        // one LDS atomic per (wave, destination present in the wave)
        uint64_t todo = m;
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int64_t dl = __shfl(d, leader, kWave);
            const uint64_t same = __ballot(ok && d == dl) & todo;
            if (lane == leader) atomicAdd(&cnt[dl], (unsigned long long)__popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= kMaxExWorld; i += blockDim.x) {
        const unsigned long long c = cnt[i];
        if (c) atomicAdd((unsigned long long *)(i == kMaxExWorld ? oob : &counts[i]), c);
    }
}
hipError_t launch_ex_dest(Src key, const uint64_t *vkey, int64_t n, int64_t pmin, int64_t pcount, int world, int64_t *dest,
                          uint64_t *vdest, int64_t *counts, int64_t *oob, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    k_ex_dest<<<grid_for(n, 256, 4), 256, 0, s>>>(key, vkey, n, pmin, pcount, world, dest, vdest, counts, oob);
    return launch_status();
}

__global__ __launch_bounds__(256) void k_ex_pack(Src src, const uint64_t *vdest, const int64_t *pos, int64_t n, int64_t *out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (bit(vdest, i)) out[pos[i]] = ld(src, i);
}
hipError_t launch_ex_pack(Src src, const uint64_t *vdest, const int64_t *pos, int64_t n, int64_t *out, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    k_ex_pack<<<grid_for(n, 256, 4), 256, 0, s>>>(src, vdest, pos, n, out);
    return launch_status();
}

// mask word per row: bit j = source vector j holds a value in that row (j < 63)
__global__ __launch_bounds__(256) void k_ex_mask(ExValid v, const uint64_t *vdest, const int64_t *pos, int64_t n, int64_t *out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!bit(vdest, i)) continue;
        uint64_t m = 0;
        for (int j = 0; j < v.n; j++) m |= (uint64_t)bit(v.valid[j], i) << j;
        out[pos[i]] = (int64_t)m;
    }
}
hipError_t launch_ex_mask(const ExValid &v, const uint64_t *vdest, const int64_t *pos, int64_t n, int64_t *out, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    k_ex_mask<<<grid_for(n, 256, 4), 256, 0, s>>>(v, vdest, pos, n, out);
    return launch_status();
}

__global__ __launch_bounds__(256) void k_ex_unmask(const int64_t *mask, int64_t n, int j, uint64_t *valid) {
    const int64_t nw = (n + 63) >> 6;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x / kWave);
    for (int64_t w = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; w < nw; w += wstride) {
        const int64_t i = (w << 6) + lane;
        const uint64_t m = __ballot(i < n && (((uint64_t)mask[i < n ? i : 0] >> j) & 1ull));
        if (lane == 0) valid[w] = m;
    }
}
hipError_t launch_ex_unmask(const int64_t *mask, int64_t n, int j, uint64_t *valid, hipStream_t s) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    k_ex_unmask<<<grid_for(n, 256, 4), 256, 0, s>>>(mask, n, j, valid);
    return launch_status();
}

}  // namespace vdl

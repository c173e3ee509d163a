// vdl_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4, wave64): the fused single-aggregate scan and
// the synthetic column generator.  (Grouped scan: vdl_mscan.hip; per-operator kernels: vdl_ops.hip; Partition and
// the row exchange: vdl_partition.hip; shared device helpers: vdl_device.h.)
//
// Everything on this path is HBM-bound integer work (no MFMA anywhere: there is no dense
// contraction in a VDL program).  The rules that matter: coalesced 16-byte-per-lane column
// loads, enough bytes in flight per CU, wave-level reductions (64-lane shuffles / ballots),
// LDS only for the cross-wave step, and one pass over every byte.
#include "vdl_device.h"

namespace vdl {

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
#ifdef VDL_SCAN_TUNING_FAMILY      // tools/build_variant.sh: the family VDL_SCAN_TUNE sweeps over (profiles/r01/tune_scan.md); not shipped
    VDL_SV(4, 1, 2, true, false, 256), VDL_SV(4, 1, 4, true, false, 256), VDL_SV(4, 1, 8, true, false, 256),
    VDL_SV(4, 1, 2, true, true, 256),  VDL_SV(4, 1, 8, true, true, 256),
    VDL_SV(4, 1, 2, true, false, 512), VDL_SV(4, 1, 4, true, false, 512), VDL_SV(4, 1, 8, true, false, 512),
    VDL_SV(4, 1, 2, true, true, 512),  VDL_SV(4, 1, 4, true, true, 512),  VDL_SV(4, 1, 8, true, true, 512),
    VDL_SV(4, 1, 10, true, true, 256), VDL_SV(4, 1, 14, true, true, 256), VDL_SV(4, 1, 12, true, false, 256), VDL_SV(4, 1, 6, true, true, 512),
    VDL_SV(4, 1, 2, true, false, 1024), VDL_SV(4, 1, 2, true, true, 1024),
#endif
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

}  // namespace vdl

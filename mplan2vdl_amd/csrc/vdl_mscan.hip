// vdl_mscan.hip -- multi-aggregate fused scans for gfx950: the global form (several Fold* over one
// filtered table: ungrouped Q1, `select sum(..), min(..), count(*) from t where ..`) and the
// grouped form (dense-domain GROUP BY: TPC-H Q1).
//
// Structure (one kernel template, GROUPED = false / true):
//   * columns: by-value descriptors, loads fully unrolled over (column, sub-iteration) exactly as in
//     k_scan -- lane l owns rows base + 512u + 2l, +1; 16-byte non-temporal loads for int64 columns;
//   * aggregates: a RUNTIME loop over descriptors in device memory (scalar loads, wave-uniform).
//     Unrolling the aggregate loop as well (the first version of this kernel) exploded into 70 K
//     instructions with thousands of SGPR spills and ran at 20 % of the HBM roofline;
//   * per aggregate the term is a product of affine column factors, evaluated for the lane's rows;
//     global form: the lane's rows are folded in registers and added to the lane's own LDS slot once
//     per tile (no atomics); grouped form: one LDS atomic per row into tab[replica][bucket][1+j];
//   * block end: 64-lane shuffle reduction + cross-wave LDS step (global) or replica fold (grouped),
//     one partial row / table per block; a second tiny kernel folds the blocks (deterministic).
// Lowering this replaces: /root/reference/src/Vlite.hs:1033-1098 (aggregates, Partition + Scatter
// + Fold per aggregate) and :721-730 (Select -> FoldSelect + Gather).
#include "vdl_kernels.h"
#include "vdl_mscan_body.h"

#include <cstdlib>
#include <cstring>

#include <map>
#include <mutex>

namespace vdl {

namespace {

static MsArgs ms_args(const MScanCols &cols) {
    MsArgs a;
    a.ncol = cols.ncol; a.n = cols.n; a.row0 = cols.row0; a.rowid_base = cols.rowid_global ? 0 : cols.row0;
    for (int c = 0; c < cols.ncol; c++) {
        a.ptr[c] = cols.ptr[c];
        a.widths |= (uint64_t)cols.width[c] << (4 * c);
        if (cols.filtered[c]) a.filtered |= 1u << c;
        if (cols.kind[c] != VC_DIRECT) a.derived |= 1u << c;
        if (cols.lazy[c]) a.lazy |= 1u << c;
    }
    return a;
}

template <int NC, int U, bool VEC, bool NT, bool GROUPED, bool DER>
__global__ __launch_bounds__(kMsBlock) void k_mscan(const MsArgs C, const MScanDesc *__restrict__ Dp) {
    mscan_body<NC, U, VEC, NT, GROUPED, DER>(C, C, *Dp, *Dp);
}

// out[w] = fold over blocks of partials[b][w]: one wave per word, lanes stride over the blocks
__global__ __launch_bounds__(256) void k_mscan_finish(const MScanDesc *__restrict__ Dp, int nblocks, int grouped, int64_t *out) {
    const MScanDesc &D = *Dp;
    const int W = D.nagg + 1;
    const int64_t words = grouped ? D.pcount * W + 1 : W;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (i >= words) return;
    int rk = R_SUM;
    if (!(grouped && i == words - 1)) {
        const int w = (int)(i % W);
        rk = w == 0 ? R_SUM : rk_of(D.agg[w - 1].kind);
    }
    int64_t x = r_identity(rk);
    for (int b = lane; b < nblocks; b += kWave) x = r_combine(rk, x, D.block_partials[(int64_t)b * words + i]);
    x = wave_reduce(x, rk);
    if (lane == 0) out[i] = x;
}

// FoldChoose per group: replace the group's smallest (global) row id by that row's column value.
// owned_only (sharded execution, after the MIN all-reduce of the row ids): only the rank that holds the
// row writes the value, every other rank writes 0, and a SUM all-reduce then spreads it.
__global__ void k_mscan_first(const MsArgs C, const MScanDesc *__restrict__ Dp, int owned_only, int64_t *table) {
    const MScanDesc &D = *Dp;
    const int W = D.nagg + 1;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= D.pcount) return;
    const bool live = table[b * W] > 0;
    for (int j = 0; j < D.nagg; j++) {
        if (D.agg[j].kind != AGG_FIRST) continue;
        int c = 0;
        for (int k = 0; k < kMaxVCols; k++) if ((D.agg[j].used >> k) & 1u) c = k;
        const int64_t r = table[b * W + 1 + j] - C.row0;
        const bool mine = live && r >= 0 && r < C.n;
        if (mine) table[b * W + 1 + j] = load_scalar(C.ptr[c], C.width(c), r);
        else if (owned_only || !live) table[b * W + 1 + j] = 0;
    }
}

// the projection scan's kernels (bodies in vdl_mscan_body.h)
template <int NC, int U, bool VEC, bool NT>
__global__ __launch_bounds__(kMsBlock) void k_project_select(const MsArgs C, const MScanDesc *__restrict__ Dp) {
    project_select_body<NC, U, VEC, NT>(C, C, *Dp, *Dp);
}
// the fused front in one pass: select-side columns / descriptor (Cs, Dsp) and take-side ones (Ct, Dtp)
template <int NCS, int NCT, int U, bool VEC, bool NT>
__global__ __launch_bounds__(kMsBlock) void k_project_front(const MsArgs Cs, const MScanDesc *__restrict__ Dsp, const MsArgs Ct, const MScanDesc *__restrict__ Dtp, const FrontLook lk) {
    project_front_body<NCS, NCT, U, VEC, NT>(Cs, Cs, *Dsp, *Dsp, Ct, Ct, *Dtp, *Dtp, lk);
}

typedef void (*mscan_fn)(const MsArgs, const MScanDesc *);
struct MsVariant { int nc, u; bool vec, grouped, der; mscan_fn fn; const char *name; };
#define VDL_MS(NC, U, VEC, NT, GR) {NC, U, VEC, GR, false, k_mscan<NC, U, VEC, NT, GR, false>, "k_mscan<" #NC "," #U "," #VEC "," #NT "," #GR ">"}
#define VDL_MSJ(NC, U, VEC, NT, GR) {NC, U, VEC, GR, true, k_mscan<NC, U, VEC, NT, GR, true>, "k_mscan_join<" #NC "," #U "," #VEC "," #NT "," #GR ">"}
const MsVariant kMsVariants[] = {
    VDL_MS(4, 6, true, true, false),  VDL_MS(8, 4, true, true, false),
    VDL_MS(4, 4, false, false, false), VDL_MS(8, 4, false, false, false),
    VDL_MS(4, 6, true, true, true),   VDL_MS(8, 2, true, true, true),   VDL_MS(8, 4, true, true, true),
    VDL_MS(4, 4, false, false, true),  VDL_MS(8, 4, false, false, true),
    VDL_MS(8, 1, true, true, true),   VDL_MS(8, 3, true, true, true),      // VDL_GROUP_U sweeps (tools/q1_ab.sh)
    // scans with derived columns (FK lookups: fused join scans)
    VDL_MSJ(8, 4, true, true, false), VDL_MSJ(8, 4, false, false, false),
    VDL_MSJ(8, 2, true, true, true),  VDL_MSJ(8, 2, false, false, true),
    VDL_MSJ(12, 2, true, true, false), VDL_MSJ(12, 2, false, false, false),
    VDL_MSJ(12, 1, true, true, true),  VDL_MSJ(12, 1, false, false, true),      // (U = 2 spills 720 B/lane in the grouped form)
};
#undef VDL_MS
#undef VDL_MSJ
constexpr int kNumMsVariants = sizeof(kMsVariants) / sizeof(kMsVariants[0]);

size_t ms_lds_bytes(const MScanDesc &d, bool grouped) {
    if (grouped) return ((size_t)((d.pcount * (d.nagg + 1)) | 1) * (size_t)d.replicas + (size_t)kMsBlock + (size_t)(d.nagg + 1)) * sizeof(int64_t);
    return (size_t)(d.nagg > 0 ? d.nagg : 1) * kMsBlock * sizeof(int64_t);
}

}  // namespace

ScanLaunch mscan_launch_config(const MScanCols &cols, MScanDesc &d, bool grouped, int num_cus) {
    bool vec = true, der = false;
    for (int c = 0; c < cols.ncol; c++) {
        if (cols.kind[c] != VC_DIRECT) { der = true; continue; }
        if (((uintptr_t)cols.ptr[c]) % (uintptr_t)(2 * cols.width[c]) != 0) vec = false;
    }
    if (cols.ncol > kMaxScanCols) der = true;              // the 12-column instantiations exist in the derived-column form only
    ScanLaunch cfg;
    cfg.variant = -1;
    const char *want_u = getenv("VDL_GROUP_U");
    for (int pass = 0; pass < 2 && cfg.variant < 0; pass++)
        for (int i = 0; i < kNumMsVariants; i++) {
            const MsVariant &v = kMsVariants[i];
            if (v.vec != vec || v.grouped != grouped || v.der != der || cols.ncol > v.nc) continue;
            if (pass == 0 && want_u && grouped && v.u != atoi(want_u)) continue;
            cfg.variant = i;
            break;
        }
    if (cfg.variant < 0) return cfg;
    d.replicas = 1;
    if (grouped) {
        d.ncomp = composite_key(d.key, d.nkey, d.comp, &d.key_masked, &d.key_mask);
        const int64_t words = d.pcount * (d.nagg + 1);
        int r = 8;
        while (r > 1 && words * r > 2304) r >>= 1;         // replicas <= 18 KiB (+ 2 KiB of trash rows): LDS never caps the occupancy
        const char *tune = getenv("VDL_GROUP_TUNE");
        if (tune) { int v = atoi(tune); if (v >= 1 && v <= 64 && words * v <= 8192) r = v; }
        d.replicas = r;
    }
    const MsVariant &v = kMsVariants[cfg.variant];
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v.fn, kMsBlock, ms_lds_bytes(d, grouped)) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 2;
    }
    if (per_cu > 8) per_cu = 8;
    const int64_t tile = (int64_t)kMsBlock * 2 * v.u;
    int64_t grid = (int64_t)num_cus * per_cu;
    if (grid > cols.n / tile) grid = cols.n / tile;
    if (grid < 1) grid = 1;
    cfg.grid = (int)grid;
    cfg.block = kMsBlock;
    return cfg;
}

const char *mscan_kernel_name(const ScanLaunch &cfg) {
    return (cfg.variant >= 0 && cfg.variant < kNumMsVariants) ? kMsVariants[cfg.variant].name : "none";
}

MsArgs mscan_args(const MScanCols &cols) { return ms_args(cols); }
size_t mscan_lds_bytes(const MScanDesc &d, bool grouped) { return ms_lds_bytes(d, grouped); }
void mscan_variant_shape(const ScanLaunch &cfg, int *nc, int *u, bool *vec, bool *grouped, bool *der) {
    const MsVariant &v = kMsVariants[cfg.variant];
    *nc = v.nc; *u = v.u; *vec = v.vec; *grouped = v.grouped; *der = v.der;
}

hipError_t launch_mscan(const MScanCols &cols, const MScanDesc &d, const MScanDesc *dev_desc, const ScanLaunch &cfg, bool grouped,
                        bool never, int64_t *out, bool resolve_first, hipStream_t s, hipFunction_t jit_fn) {
    (void)hipGetLastError();
    if (cfg.variant < 0 || cfg.variant >= kNumMsVariants) return hipErrorInvalidValue;
    int nblocks = 0;
    if (!never && cols.n > 0) {
        if (jit_fn) {
            MsArgs a = ms_args(cols);
            void *params[] = {&a, &dev_desc};
            const hipError_t e = hipModuleLaunchKernel(jit_fn, (unsigned)cfg.grid, 1, 1, (unsigned)cfg.block, 1, 1, (unsigned)ms_lds_bytes(d, grouped), s, params, nullptr);
            if (e != hipSuccess) return e;
        } else {
            hipLaunchKernelGGL(kMsVariants[cfg.variant].fn, dim3(cfg.grid), dim3(cfg.block), ms_lds_bytes(d, grouped), s, ms_args(cols), dev_desc);
        }
        nblocks = cfg.grid;
    }
    const int64_t words = grouped ? d.pcount * (d.nagg + 1) + 1 : d.nagg + 1;
    k_mscan_finish<<<(int)((words + 3) / 4), 256, 0, s>>>(dev_desc, nblocks, grouped ? 1 : 0, out);
    if (grouped && resolve_first) k_mscan_first<<<(int)((d.pcount + 255) / 256), 256, 0, s>>>(ms_args(cols), dev_desc, 0, out);
    return hipGetLastError();
}

int64_t project_tiles(int64_t n) { return (n + kProjTile - 1) / kProjTile; }

bool project_select_vec(const MScanCols &cols) {
    bool vec = true;
    for (int c = 0; c < cols.ncol; c++)
        if (cols.kind[c] == VC_DIRECT && !cols.lazy[c] && ((uintptr_t)cols.ptr[c]) % (uintptr_t)(2 * cols.width[c]) != 0) vec = false;
    return vec;
}
// blocks per CU of a projection kernel: by the kernel's registers when it was specialised, else by its column count
static int project_blocks_per_cu(hipFunction_t jit_fn, int ncol) {
    int per_cu = 8;
    if (jit_fn) {
        static std::mutex mu;
        static std::map<hipFunction_t, int> known;
        std::lock_guard<std::mutex> g(mu);
        auto it = known.find(jit_fn);
        if (it == known.end()) {
            int n = 0;
            if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&n, jit_fn, kMsBlock, 0) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 8; }
            if (known.size() > 4096) known.clear();
            it = known.emplace(jit_fn, n).first;
        }
        per_cu = it->second;
    } else {
        per_cu = ncol <= 4 ? 8 : ncol <= 8 ? 5 : 3;            // (precompiled: 4 / 8 / 12 columns x 8 rows of 64 bits in registers)
    }
    if (const char *e = getenv("VDL_PROJ_BLOCKS_PER_CU")) { if (atoi(e) > 0) per_cu = atoi(e); }
    return per_cu;
}
hipError_t launch_project_select(const MScanCols &cols, const MScanDesc *dev_desc, int num_cus, hipStream_t s, hipFunction_t jit_fn) {
    (void)hipGetLastError();
    if (cols.n <= 0) return hipSuccess;
    const bool vec = project_select_vec(cols);
    // The blocks walk the tiles with the grid as their stride, so the grid is what the chip holds at once -- blocks per CU by the
    // kernel's registers: no block waits for a slot while others hold theirs for the whole pass.
    const int per_cu = project_blocks_per_cu(jit_fn, cols.ncol);
    int64_t grid = project_tiles(cols.n);
    if (grid > (int64_t)num_cus * per_cu) grid = (int64_t)num_cus * per_cu;
    MsArgs a = ms_args(cols);
    if (jit_fn) {
        void *params[] = {&a, &dev_desc};
        return hipModuleLaunchKernel(jit_fn, (unsigned)grid, 1, 1, kMsBlock, 1, 1, 0, s, params, nullptr);
    }
#define VDL_PJ(NC) do { if (vec) k_project_select<NC, kProjU, true, true><<<(int)grid, kMsBlock, 0, s>>>(a, dev_desc); \
                        else k_project_select<NC, kProjU, false, false><<<(int)grid, kMsBlock, 0, s>>>(a, dev_desc); } while (0)
    if (cols.ncol > kMaxSelectCols) return hipErrorInvalidValue;
    if (cols.ncol <= 4) VDL_PJ(4); else if (cols.ncol <= 8) VDL_PJ(8); else VDL_PJ(kMaxSelectCols);      // (registers: NC x 8 rows x 64 bits)
#undef VDL_PJ
    return hipGetLastError();
}
int64_t project_look_bytes(int64_t n) { return 64 + (int64_t)sizeof(unsigned long long) * front_look_words((project_tiles(n) + kFrontBatch - 1) / kFrontBatch); }
// look: project_look_bytes(n) bytes, zeroed here; total_dev / total_host: where the survivors' number is left
hipError_t launch_project_front(const MScanCols &scols, const MScanDesc *dev_sdesc, const MScanCols &tcols, const MScanDesc *dev_tdesc, void *look,
                                int64_t *total_dev, int64_t *total_host, int num_cus, hipStream_t s, hipFunction_t jit_fn) {
    (void)hipGetLastError();
    if (scols.n <= 0) return hipSuccess;
    if (project_tiles(scols.n) >= ((int64_t)1 << (kFrontFanBits * kFrontLevels))) return hipErrorInvalidValue;
    // (one kernel: the runtime's memset of a size like this one came as two fill kernels of 5 us each)
    hipError_t e = launch_fill_words((uint64_t *)look, 0, project_look_bytes(scols.n) / (int64_t)sizeof(uint64_t), s);
    if (e != hipSuccess) return e;
    FrontLook lk;
    lk.ticket = (unsigned int *)look; lk.nodes = (unsigned long long *)((char *)look + 64); lk.total = total_dev; lk.total_host = total_host;
    const bool vec = project_select_vec(scols);
    int64_t grid = (project_tiles(scols.n) + kFrontBatch - 1) / kFrontBatch;
    const int per_cu = project_blocks_per_cu(jit_fn, scols.ncol);
    if (grid > (int64_t)num_cus * per_cu) grid = (int64_t)num_cus * per_cu;        // (tiles are handed out by ticket: any grid is safe, this one fills the chip)
    MsArgs a = ms_args(scols), b = ms_args(tcols);
    if (jit_fn) {
        void *params[] = {&a, &dev_sdesc, &b, &dev_tdesc, &lk};
        return hipModuleLaunchKernel(jit_fn, (unsigned)grid, 1, 1, kMsBlock, 1, 1, 0, s, params, nullptr);
    }
    if (scols.ncol > kMaxSelectCols) return hipErrorInvalidValue;
#define VDL_PF2(NCS, NCT) do { if (vec) k_project_front<NCS, NCT, kProjU, true, true><<<(int)grid, kMsBlock, 0, s>>>(a, dev_sdesc, b, dev_tdesc, lk); \
                               else k_project_front<NCS, NCT, kProjU, false, false><<<(int)grid, kMsBlock, 0, s>>>(a, dev_sdesc, b, dev_tdesc, lk); } while (0)
#define VDL_PF(NCS) do { if (tcols.ncol <= 12) VDL_PF2(NCS, 12); else VDL_PF2(NCS, kMaxVCols); } while (0)
    if (scols.ncol <= 4) VDL_PF(4); else if (scols.ncol <= 8) VDL_PF(8); else VDL_PF(kMaxSelectCols);
#undef VDL_PF
#undef VDL_PF2
    return hipGetLastError();
}
hipError_t launch_mscan_resolve_first(const MScanCols &cols, const MScanDesc &d, const MScanDesc *dev_desc, int64_t *table, hipStream_t s) {
    (void)hipGetLastError();
    k_mscan_first<<<(int)((d.pcount + 255) / 256), 256, 0, s>>>(ms_args(cols), dev_desc, 1, table);
    return hipGetLastError();
}

}  // namespace vdl

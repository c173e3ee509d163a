// vdl_engine_internal.h -- what the translation units behind the C ABI share: the HBM pool, the column catalog,
// the vector forms of the per-operator executor, and the definitions of vdl_ctx / vdl_plan.  Not installed;
// include/vdl.h is the interface.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "vdl.h"
#include "vdl_exchange_analysis.h"
#include "vdl_fuse.h"
#include "vdl_ir.h"
#include "vdl_jit.h"
#include "vdl_kernels.h"


namespace vdl {
namespace eng {


#define HIP_CHECK(expr)                                                                                         \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess)                                                                                   \
            throw Error(VDL_ERR_DEVICE, std::string(#expr) + " failed: " + hipGetErrorString(e_));              \
    } while (0)

// ---- HBM pool: size-class free lists; everything runs on one stream, so a buffer released on the
// host can be handed to a later launch without extra synchronisation (stream order protects it).
struct Pool {
    std::multimap<size_t, void *> free_list;
    size_t live_bytes = 0, peak_bytes = 0;
    static size_t round_up(size_t b) {
        size_t c = 256;
        while (c < b) c <<= 1;
        if (c > (size_t(1) << 26)) c = (b + (size_t(1) << 26) - 1) & ~((size_t(1) << 26) - 1);   // 64 MiB granules above 64 MiB
        return c;
    }
    void *alloc(size_t bytes, size_t *cls) {
        size_t c = round_up(bytes ? bytes : 1);
        *cls = c;
        auto it = free_list.find(c);
        void *p = nullptr;
        if (it != free_list.end()) { p = it->second; free_list.erase(it); }
        else {
            hipError_t e = hipMalloc(&p, c);
            if (e != hipSuccess) {
                trim();
                e = hipMalloc(&p, c);
                if (e != hipSuccess) throw Error(VDL_ERR_NOMEM, "hipMalloc of " + std::to_string(c) + " bytes failed");
            }
        }
        live_bytes += c;
        peak_bytes = std::max(peak_bytes, live_bytes);
        return p;
    }
    void release(void *p, size_t cls) {
        if (closed) { (void)hipFree(p); return; }        // the context is gone: give the memory back at once
        free_list.emplace(cls, p);
        live_bytes -= cls;
    }
    void trim() { for (auto &kv : free_list) (void)hipFree(kv.second); free_list.clear(); }
    size_t idle_bytes() const { size_t b = 0; for (const auto &kv : free_list) b += kv.first; return b; }     // parked buffers: handed back to the device when an allocation fails
    bool closed = false;
    ~Pool() { trim(); }
};

struct DevBuf {
    void *p = nullptr;
    size_t cls = 0;
    std::shared_ptr<Pool> pool;       // buffers (held by plans) may outlive their context
    ~DevBuf() { if (p && pool) pool->release(p, cls); }
};
using BufP = std::shared_ptr<DevBuf>;

struct Column {
    const void *dev = nullptr;
    int width = 0;
    int64_t n = 0;
    BufP owned;
};

// device-side vector of the general path
// A selection: m of n slots, ascending.  Vectors that hold values only on a sparse selection (after a selective
// filter, Vlite.hs:721-730) are stored as SPARSE: the m values of the selected slots, so that everything
// downstream of the filter touches m instead of n elements (GenExec: "sparse vectors").
struct Sel {
    int64_t n = 0, m = 0;
    BufP idx;                       // the m slot ids; null = the prefix 0 .. m-1
    BufP bitmap;                    // n bits with exactly the selected slots set (prefix selections: built on demand)
    std::shared_ptr<Sel> parent;    // the selection this one was filtered from, and
    BufP ppos;                      // for each of the m slots its entry number inside the parent
    bool worth = true;              // false: too dense to be worth compacting (only m is known)
    BufP wrank;                     // selected slots before each bitmap word (built when something gathers out of a vector on this selection)
    int64_t first_slot = -1;        // the first selected slot once somebody asked (GenExec::first_slot_of); prefix selections: 0
};
using SelP = std::shared_ptr<Sel>;

struct ExprNode;
struct LazyGather;
struct CommState;        // vdl_comm.cpp: the context's communicator (RCCL or host transport)
struct ShardState;       // vdl_comm.cpp: per-plan buffers of the sharded fold route

struct DVec {
    enum Kind { NONE, DENSE, COLUMN, RANGE, ONEHOT, OHCONST, SPARSE, EXPR, LAZYG } kind = NONE;
    int64_t n = 0;
    SelP sel;                   // SPARSE: data = the sel->m values; valid = bitmap over those m entries (null = all hold a value)
    bool perm = false;          // SPARSE: the values are a permutation of 0 .. m-1 (Partition positions)
    bool iota = false;          // ... and that permutation is the identity (the partitioned data was already in order)
    bool ranks = false;         // DENSE Partition positions: the values of the valid slots are exactly 0 .. m-1 (m = number of valid slots)
    bool ids = false;           // SPARSE: every value is its own slot id (row ids gathered through a filter)
    std::shared_ptr<LazyGather> lg; // LAZYG: Gather(src, pos) not run yet: its only reader is a filter that needs few (or none) of its values
    std::shared_ptr<ExprNode> ex;   // EXPR: a not yet evaluated tree of element-wise operators (fused when somebody needs the values)
    BufP data;                  // DENSE: n int64; ONEHOT/OHCONST: {value, slot, count}
    const void *ptr = nullptr;  // COLUMN: borrowed catalog pointer
    int width = 8;
    int64_t from = 0, step = 0; // RANGE; OHCONST: from = the constant
    BufP valid;                 // bitmap, null = every slot holds a value
    BufP keep;                  // COLUMN: keeps an engine-owned column alive
    BufP sorted_keys, sorted_keys_src;   // with `order`: the partitioned values themselves in rank order, and the buffer they are the values of
                                // (a Scatter of that very vector by these positions -- the key of a GROUP BY -- is then already written)
    BufP order;                 // Partition positions whose only readers are Scatters: the slots in RANK order (order[pos[slot]] = slot); `data` (the
                                // positions themselves) is filled in only if somebody asks (GenExec::need_positions)
};

// Element-wise operators whose only reader is another element-wise operator are not run one by one: they pile up
// in a tree whose leaves are stored vectors, and the tree runs as one kernel (k_expr) when its root is needed.
struct ExprNode {
    int bin = -1;                          // -1: leaf
    std::shared_ptr<ExprNode> l, r;
    DVec leaf;                             // DENSE / COLUMN / RANGE
    int leaves = 1, instrs = 1, depth = 1;
};

struct LazyGather { DVec src, pos; };        // both in dense form (DENSE / COLUMN / RANGE)

constexpr size_t kBigOutput = 1u << 16;     // values; from here on results use pinned host memory (or stay on the device)
struct Output {
    int node = 0;
    std::string name, tmp;
    std::vector<int64_t> vals;
    const int64_t *big = nullptr;      // large results land in a pinned buffer the plan keeps (pageable copies run at a few GB/s)
    size_t big_n = 0;
    std::shared_ptr<void> dev_keep;    // vdl_plan_set_device_outputs: large results stay in HBM, owned by the plan until its next run
    const int64_t *dev = nullptr;
    const int64_t *ptr() const { return dev ? nullptr : big ? big : vals.data(); }
    size_t count() const { return (dev || big) ? big_n : vals.size(); }
};
struct Timing { std::string label; double usec; };
// vdl_plan_set_trace: a host copy of a statement's vector in semantic form (n slots: value + "holds a value"), taken
// right after the statement ran on the per-operator executor.  `have` is false when the vector was not evaluated at
// that point (pending expression tree / lazy gather) or is longer than kTraceMaxSlots.
constexpr int64_t kTraceMaxSlots = (int64_t)1 << 22;
struct Traced { int node = 0; const char *form = ""; int64_t n = 0; bool have = false; std::vector<int64_t> vals; std::vector<uint8_t> ok; };


}  // namespace eng
}  // namespace vdl

using namespace vdl;
using namespace vdl::eng;


struct vdl_ctx {
    int device = -1;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t copy_stream = nullptr;     // result copies of the general path run here, behind an event, while later statements compute
    hipEvent_t copy_ev = nullptr;
    int num_cus = 256;
    std::string arch = "gfx950";           // --offload-arch of run-time specialisation (the device's, when there is one)
    std::map<std::string, Column> cols;
    uint64_t catalog_version = 1;      // bumped on every catalog change: plans re-bind only when it moved
    const std::map<std::string, Column> *overlay = nullptr;     // columns that stand in for catalog entries during one run (vdl_comm.cpp: sharded_replicate)
    uint64_t overlay_epoch = 0;                                  // moves whenever the overlay is set or cleared
    // what bindings and kernels specialised for column addresses / widths / lengths are keyed by: the catalog's state AND the overlay's
    uint64_t binding_version() const { return catalog_version + (overlay_epoch << 40); }
    std::shared_ptr<Pool> pool = std::make_shared<Pool>();
    std::shared_ptr<CommState> comm;       // vdl_comm_init / vdl_comm_init_host
    std::string err;
    // a few words of PINNED host memory for the round trips of the executors (survivor counts, sortedness verdicts): a copy
    // into pageable memory is staged by the runtime and cost 20-30 us of idle GPU each (Q3 at SF10: three of them per query)
    int64_t *pinned_words = nullptr;
    static constexpr int kPinnedWords = 1 << 17;       // 1 MiB: also the tables the sharded routes gather (the exchange's: ranks x (4096-slice histogram + a few words) -- eight ranks' did not fit the 256 KiB of round 3 by 32 words and took the staged copy)
    static constexpr int kFlagWords = 8;                // the last words: [0] the flag a posting kernel raises, [1] the fused front's survivor count
    int64_t *pinned(int64_t words) {
        if (!pinned_words && hipHostMalloc((void **)&pinned_words, sizeof(int64_t) * kPinnedWords, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); pinned_words = nullptr; }
        return words <= kPinnedWords - kFlagWords ? pinned_words : nullptr;
    }
    int64_t *flag_words() { return pinned(1) ? pinned_words + (kPinnedWords - kFlagWords) : nullptr; }
    BufP sorted_state;                 // four device words the fused sortedness pass keeps between its launches (k_sorted_heads_counted)
    int64_t post_seq = 0;
    // Round trips without hipStreamSynchronize: a one-block kernel behind whatever is queued posts words and a sequence number into
    // pinned memory with system-scope stores, the host polls the number (and the stream's state now and then, so that a failed
    // launch ends the wait).  wait_here: "everything queued on s so far is done".
    void wait_flag(int64_t *word, int64_t until_not, hipStream_t s);          // until *word != until_not
    // until *word == seq.  (The word is read ONCE per turn: "while (*w != seq) wait_flag(w, *w, s)" read it twice, and a post that landed
    // between the two reads left wait_flag waiting for the word to move away from `seq` -- until its idle-stream check, a few
    // milliseconds later, declared the kernel lost: one spurious "did not report" in some thousands of round trips.)
    void wait_seq(int64_t *word, int64_t seq, hipStream_t s) {
        for (;;) {
            const int64_t seen = *(volatile int64_t *)word;
            if (seen == seq) break;
            wait_flag(word, seen, s);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    void wait_here(hipStream_t s);
    // `k` int64 words from the device to `out`, waited for on `s`: through the pinned words when they fit -- no copy of this library
    // has pageable host memory for its destination unless it is larger than that (result vectors, traces)
    void fetch_to_host(const void *dev, size_t k, int64_t *out, hipStream_t s);
    // pinned staging for small result vectors (GenExec::copy_out): copied in stream order, read after ONE synchronise per run
    static constexpr size_t kSmallStageWords = (size_t)1 << 17;
    int64_t *small_stage_words = nullptr;
    int64_t *small_stage() {
        if (!small_stage_words && hipHostMalloc((void **)&small_stage_words, sizeof(int64_t) * kSmallStageWords, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); small_stage_words = nullptr; }
        return small_stage_words;
    }
    ~vdl_ctx() { if (pinned_words) (void)hipHostFree(pinned_words); if (small_stage_words) (void)hipHostFree(small_stage_words); }
};

struct vdl_plan {
    vdl_ctx *ctx = nullptr;          // only dereferenced inside calls that receive the live context
    int device = -1;                 // copied at parse time: the plan may outlive its context
    Program prog;
    FusedPlan fused;
    bool use_fusion = true;
    bool profiling = false;
    bool tracing = false;
    std::vector<Traced> traced;
    hipEvent_t stmt_ev[2] = {nullptr, nullptr};   // per-statement profiling of the per-operator executor (created on first use)
    bool device_outputs = false;
    std::string description;
    std::vector<Output> outs;
    std::vector<Timing> timings;
    // fused state
    std::vector<ScanArgs> sargs;
    std::vector<ScanLaunch> scfg;
    std::vector<BufP> block_partials;
    std::vector<int32_t> reduce_ops;
    std::vector<int64_t> word_offset;
    std::vector<MScanCols> mcols;            // [scans..., gscans...] entries that run on k_mscan
    std::vector<MScanDesc> mdesc;
    std::vector<ScanLaunch> mcfg;
    std::vector<BufP> mparts, mdev;
    bool use_jit = false;                    // vdl_plan_set_jit / VDL_JIT=1: scans specialised for this plan by hiprtc (vdl_jit.cpp)
    bool jit_tune = false, jit_tuned = false;   // ... =2: rows per lane chosen by timing at the first run
    std::vector<std::shared_ptr<vdl::jit::Kernel>> mjit;
    struct JitForm { int u = 0, lazy = 0; };     // rows per lane / filter columns read with the tile (0 = not staged) of mjit[s]
    std::vector<JitForm> mjit_form;
    struct FrontKernel { uint64_t version = 0; std::shared_ptr<vdl::jit::Kernel> k; };
    std::map<std::string, FrontKernel> front_jit;   // specialised passes of the projection scan / dimension scans, by role
    std::vector<std::shared_ptr<vdl::MScanDesc>> host_descs;     // dimension scans of the current run (copied to the device asynchronously)
    std::shared_ptr<void> front_keep;                        // the fused front's bound descriptors of the current run, likewise
    std::vector<char> kscan;                 // [scan] runs on the single-aggregate k_scan (decided when the plan is bound / tuned)
    std::string jit_note;                    // what was specialised, or why not
    std::vector<BufP> prelude_buf;           // fused join scans: dimension bitmaps / LIKE tables of the current run (FusedPlan::prelude)
    std::vector<int64_t> prelude_n;
    std::vector<int64_t> prelude_rows;       // SEMI_BITMAP items: rows of the (local) source table the set was built from
    // sharded runs of a plan with a semi-join set (vdl_comm.cpp): every rank builds the set from its rows of the source table,
    // `after_prelude` merges the ranks' sets (and clips them at the GLOBAL length of that table) before any scan reads them
    bool semi_unclamped = false;
    std::map<std::string, Column> replica;   // "replicate" route: the sharded table's columns this plan loads, all ranks' rows (gathered when the catalog changed)
    uint64_t replica_version = 0;
    bool front_rowid_global = false;         // sharded front route: the front's row-id columns are global row numbers
    std::function<void(vdl_ctx *, vdl_plan *)> after_prelude;
    // sharded "front" route (vdl_comm.cpp): called once the fused front has run over this rank's rows -- or has failed, or was
    // abandoned: `failure` says why -- to replace the front's vectors by the ranks' vectors one after the other
    std::function<void(vdl_ctx *, vdl_plan *, std::map<int, DVec> &, bool, const std::string &)> after_front;
    std::vector<int64_t> gword_offset;
    int dominant = -1;
    std::string dominant_kernel;
    std::string traffic_detail;            // vdl_plan_scan_traffic: per-column bytes of the last census
    int64_t n_words = 0;
    int64_t row_offset = 0;                  // global index of this rank's first row (sharded FoldChoose)
    // sharded Partition exchange (vdl_exchange_*)
    struct ExState {
        bool active = false;
        int world = 0;
        int64_t n = 0, n_send = 0;
        std::vector<DVec> src;             // [0] = key, then the other scattered vectors
        BufP tileoff, owner, routecnt;     // where the rows of every destination begin, tile by tile; the slice -> rank table (device); launch_ex_route's counts
        ExRoute route;                     // how a row's destination is found (exchange_route; the key lives in src[0])
        bool skip_mask = false;            // vdl_run_sharded: no rank's vectors have holes, so the mask column is neither written nor sent nor read
        std::vector<int> nodes;
        int64_t pmin = 0, pcount = 0;
        // global folds over the sharded table that the tail reads beside the Partition (Q11: HAVING sum(..) > (select sum(..) * k)):
        // their local records travel with the counts as three mergeable words each {value, first global row, count}
        std::vector<int> folds;
        std::vector<int64_t> fold_n, fold_words, fold_merged;
    } ex;
    // "chain" route: stage 1 = exchange at the first Partition with the sets' operands as targets (the packed positions of every set are
    // left in `lists`); stage 2 = the rest of the program with the merged `sets` in place of their statements
    struct ChainRun {
        int stage = 0;
        std::shared_ptr<ChainPlan> plan;
        std::string plan_table, why;       // the placement `plan` was analysed for; why there is none
        struct SetList { BufP list; int64_t m = 0, len = 0; };
        std::vector<SetList> lists;
        std::map<int, DVec> sets;
    } chain;
    bool ex_allow_folds = false;           // set by vdl_run_sharded around the exchange calls (callers of the bare calls get no fold merge)
    std::shared_ptr<ShardState> shard;     // vdl_run_sharded: send / receive / merged word buffers
    BufP shard_keep;                       // vdl_run_sharded: received rows while the tail of an exchange plan reads them
    std::string sharded_table;             // placement named in the last vdl_exchange_spec ("" = not stated)
    std::vector<int> cut_folds;            // general plan sharded through its global folds: the folds of the last vdl_run_local
    std::vector<int64_t> cut_n;            // and the lengths of their operands on this rank
    BufP words;
    int64_t words_cap = 0;
    std::string fallback_note, front_note;
    // scan descriptors on the device, by role ("scan3", "dim0", "select", "take"): a plan-owned buffer and the bytes it holds -- the
    // descriptor of a run is uploaded only when it differs from what is there (pool buffers come back at the same addresses run
    // after run, so it rarely does: each upload was a 5 us staged copy on the stream plus the host's part of it)
    struct DescSlot { BufP dev; std::vector<unsigned char> shadow; uint64_t used = 0; };
    std::map<std::string, DescSlot> desc_slots;
    uint64_t desc_clock = 0;
    double front_usec = 0;
    int64_t front_m_seen = -1;               // survivors of the front's last run: the next run launches its take pass with room for about as many
    bool bound = false;
    uint64_t bound_version = 0;
    // pipelined finalisation: two pinned host slots, one event each
    int64_t *host_words[2] = {nullptr, nullptr};
    int64_t host_cap = 0;
    hipEvent_t slot_ev[2] = {nullptr, nullptr};
    bool slot_pending[2] = {false, false};
    int64_t scan_rows = 0, scan_bytes = 0;
    double scan_usec = 0;
    static constexpr int kEvRing = 4;   // runs in flight before their timing is read (pipelined callers: up to 3)
    hipEvent_t ev0[kEvRing] = {}, ev1[kEvRing] = {};   // profiling events, one pair per run, ring
    bool ev_pending[kEvRing] = {}, ev_bound[kEvRing] = {};
    uint64_t ev_seq[kEvRing] = {};
    unsigned run_seq = 0;
    int last_ev = 0;
    const void *ev_buf[kEvRing] = {};   // partial-word buffer each event pair's run wrote (pipelined callers finalise out of order)
    int slot_ev_idx[2] = {-1, -1};
    // pinned result buffers of the general path, one per output ordinal, grown on demand
    std::vector<std::pair<int64_t *, size_t>> out_pinned;
    int64_t *pinned_out(size_t ordinal, size_t count) {
        if (out_pinned.size() <= ordinal) out_pinned.resize(ordinal + 1, {nullptr, 0});
        auto &b = out_pinned[ordinal];
        if (b.second < count) {
            if (b.first) (void)hipHostFree(b.first);
            b.first = nullptr; b.second = 0;
            if (hipHostMalloc((void **)&b.first, sizeof(int64_t) * count, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); b.first = nullptr; return nullptr; }
            b.second = count;
        }
        return b.first;
    }
    ~vdl_plan() {
        for (auto &b : out_pinned) if (b.first) (void)hipHostFree(b.first);
        for (hipEvent_t e : stmt_ev) if (e) (void)hipEventDestroy(e);
        for (int k = 0; k < 2; k++) {
            if (ev0[k]) (void)hipEventDestroy(ev0[k]);
            if (ev1[k]) (void)hipEventDestroy(ev1[k]);
            if (ev0[k + 2]) (void)hipEventDestroy(ev0[k + 2]);
            if (ev1[k + 2]) (void)hipEventDestroy(ev1[k + 2]);
            if (slot_ev[k]) (void)hipEventDestroy(slot_ev[k]);
            if (host_words[k]) (void)hipHostFree(host_words[k]);
        }
    }
};

namespace vdl {
namespace eng {


// the sharded Partition's local phase in steps (vdl_exchange.cpp): vdl_exchange_begin = local + route(null); vdl_run_sharded puts the
// ranks' key histograms in between and routes by the cut they give
void exchange_local(vdl_ctx *c, vdl_plan *p, int world);
bool exchange_has_holes(const vdl_plan *p);                                        // after exchange_local: some travelling vector but the key has empty slots of its own
std::shared_ptr<ChainPlan> chain_plan(vdl_plan *p, std::string &why);             // the plan's chain route for p->sharded_table, or null and why not
void chain_build_set(vdl_ctx *c, vdl_plan *p, size_t k, const BufP &positions, int64_t m, int64_t len);   // p->chain.sets[set k] from everybody's positions
void chain_run_everywhere(vdl_ctx *c, vdl_plan *p);                               // stage 2 without a second cut: the rest on every rank
void exchange_histogram(vdl_ctx *c, vdl_plan *p, int64_t *hist_host /* kExBins + 1 */);
void exchange_route(vdl_ctx *c, vdl_plan *p, const int32_t *owner_host /* kExBins, or null */, int64_t *counts_host);

inline BufP dev_alloc(vdl_ctx *c, size_t bytes) {
    auto b = std::make_shared<DevBuf>();
    b->pool = c->pool;
    b->p = c->pool->alloc(bytes, &b->cls);
    // debugging aid: VDL_POISON=<byte> fills every buffer handed out, so that a kernel reading what nobody wrote fails the
    // same way every time instead of depending on what the memory held before; VDL_POISON=rand[<seed>] fills it with
    // plausible garbage instead (k_poison), a different mix per buffer
    static const char *poison = getenv("VDL_POISON");
    if (poison && b->p) {
        if (!std::strncmp(poison, "rand", 4)) {
            static uint64_t counter = 0;
            (void)launch_poison(b->p, b->cls, (uint64_t)std::strtoull(poison + 4, nullptr, 10) * 1000003ull + 8 * (++counter), c->stream);
        } else (void)hipMemsetAsync(b->p, atoi(poison) & 255, b->cls, c->stream);
    }
    return b;
}

inline void need_device(vdl_ctx *c) {
    if (c->device < 0) throw Error(VDL_ERR_DEVICE, "this context has no HIP device (opened with device < 0)");
    HIP_CHECK(hipSetDevice(c->device));
}

inline uint64_t fnv1a(const std::string &s) {
    uint64_t h = 0xCBF29CE484222325ULL;
    for (unsigned char ch : s) { h ^= ch; h *= 0x100000001B3ULL; }
    return h;
}

inline const Column &find_col(vdl_ctx *c, const std::string &name) {
    if (c->overlay) {                                   // a sharded run on the "replicate" route: the whole table instead of this rank's rows
        auto o = c->overlay->find(name);
        if (o != c->overlay->end()) return o->second;
    }
    auto it = c->cols.find(name);
    if (it == c->cols.end()) throw Error(VDL_ERR_COLUMN, "Load: column '" + name + "' is not in the catalog");
    return it->second;
}

std::string describe_plan(const vdl_plan *p);
size_t exchange_fold_count(const vdl_plan *p, const std::string &table);      // vdl_exchange.cpp: global folds beside the Partition (sharded runs)
int exchange_fold_kind(const vdl_plan *p, size_t k);                          // ... 0 sum / count, 1 min, 2 max of the k-th one (after vdl_exchange_begin)
bool run_projection(vdl_ctx *c, vdl_plan *p, std::map<int, DVec> &over);     // vdl_engine.cpp: the fused front of a plan that does not fuse as a whole
// general (not fused) plans sharded by rows through their global folds, vdl_exchange.cpp
bool general_partial_spec(const vdl_plan *p, std::vector<int32_t> &ops, std::string &why);
void general_run_local(vdl_ctx *c, vdl_plan *p, int64_t *dev_words);
void general_finalize(vdl_ctx *c, vdl_plan *p, const int64_t *dev_words);

template <typename F>
int guard(vdl_ctx *c, F &&f) {
    try {
        f();
        return VDL_OK;
    } catch (const Error &e) {
        if (c) c->err = e.what();
        return e.code;
    } catch (const std::bad_alloc &) {
        if (c) c->err = "out of host memory";
        return VDL_ERR_NOMEM;
    } catch (const std::exception &e) {
        if (c) c->err = e.what();
        return VDL_ERR_ARG;
    }
}


}  // namespace eng
}  // namespace vdl

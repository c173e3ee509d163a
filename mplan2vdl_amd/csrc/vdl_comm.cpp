// vdl_comm.cpp -- the multi-GPU path behind the C ABI (SURVEY.md section 8(b),(e)): one process per GPU, a context owns its
// communicator, vdl_run_sharded() runs a plan over this rank's rows and performs the collectives itself.
//
//   transport RCCL : librccl is opened at vdl_comm_init (dlopen: single-GPU users and CPU boxes need no RCCL); collectives are
//                    queued on a communication stream of the context, ordered against the engine stream with events, so the
//                    merge of query k overlaps the scan of query k+1 (vdl_run_sharded_begin / _end).
//   transport HOST : the caller supplies all-gather / all-to-all over HOST memory (vdl_comm_init_host: MPI or gloo hosts,
//                    and the in-process stand-in of the tests); the engine stages through pinned buffers.
//
// Collectives per query -- what bounds 8-GPU scaling of a 0.3 ms scan is their count and latency, not their size:
//   plans whose outputs are global / dense-domain grouped folds (Q6, Q1; Q14, Q19 through their fold records):
//       ONE all-gather of the partial words (Q6: 2 words, Q1: 289) + a local merge kernel that applies each word's
//       VDL_REDUCE_* operator over the ranks.  FoldChoose words travel as (global row id, value) pairs, so the
//       MIN -> resolve -> SUM double round of the all-reduce formulation is gone.
//   plans with a Partition (Q3, Q5, Q9, Q10, Q12, Q20):
//       ONE all-gather of {status, rows per destination} (the only host synchronisation: receive sizes depend on it), then
//       ONE grouped send/receive (ncclGroupStart .. ncclGroupEnd) that moves every column's piece for every peer.
#include <dlfcn.h>

#include "vdl_engine_internal.h"

namespace vdl {
namespace eng {

// ---- the slice of the RCCL API this file uses (types as in rccl.h; the library is bound at run time) ----
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
static_assert(sizeof(ncclUniqueId) == VDL_COMM_ID_BYTES, "vdl.h promises the size of ncclUniqueId");
enum { kNcclSuccess = 0, kNcclInt8 = 0, kNcclInt64 = 4 };          // ncclResult_t / ncclDataType_t values (nccl.h: ncclInt8 = 0, ncclInt64 = 4)
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(const ncclComm_t, int *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

static Rccl &rccl() {
    static Rccl r;
    if (r.lib) return r;
    const char *names[] = {getenv("VDL_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string tried;
    for (const char *n : names) {
        if (!n || !*n) continue;
        r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
        tried += std::string(" ") + n + " (" + dlerror() + ")";
    }
    if (!r.lib) throw Error(VDL_ERR_DEVICE, "cannot load RCCL:" + tried);
    auto sym = [&](const char *name) {
        void *p = dlsym(r.lib, name);
        if (!p) { std::string m = std::string("RCCL library lacks ") + name; dlclose(r.lib); r.lib = nullptr; throw Error(VDL_ERR_DEVICE, m); }
        return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    return r;
}
#define RCCL_CHECK(expr)                                                                                               \
    do {                                                                                                               \
        int e_ = (expr);                                                                                               \
        if (e_ != kNcclSuccess) throw Error(VDL_ERR_DEVICE, std::string(#expr) + " failed: " + rccl().GetErrorString(e_)); \
    } while (0)

struct CommState {
    enum Kind { RCCL, HOST } kind = HOST;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    vdl_comm_host host{};
    hipStream_t stream = nullptr;               // communication stream (RCCL transport)
    hipEvent_t ev_local[2] = {nullptr, nullptr};    // engine stream -> communication stream: the partial words of slot k are written
    hipEvent_t ev_merged[2] = {nullptr, nullptr};   // communication stream: the send buffer of slot k has been consumed
    char *stage = nullptr;                      // pinned staging for the HOST transport
    size_t stage_cap = 0;
    ~CommState() {
        if (comm) (void)rccl().CommDestroy(comm);
        for (int k = 0; k < 2; k++) {
            if (ev_local[k]) (void)hipEventDestroy(ev_local[k]);
            if (ev_merged[k]) (void)hipEventDestroy(ev_merged[k]);
        }
        if (stream) (void)hipStreamDestroy(stream);
        if (stage) (void)hipHostFree(stage);
    }
    char *staging(size_t bytes) {
        if (stage_cap < bytes) {
            if (stage) (void)hipHostFree(stage);
            stage = nullptr; stage_cap = 0;
            HIP_CHECK(hipHostMalloc((void **)&stage, bytes, hipHostMallocDefault));
            stage_cap = bytes;
        }
        return stage;
    }
};

// per-plan buffers of the sharded fold route (two slots: pipelined callers)
// A rank's block in the all-gather is 2 * n_words + 1 int64: the words raw, the words with FoldChoose entries resolved, and the
// STATUS of its local phase (VDL_OK or the error code).  A rank whose local phase failed still takes part in the collective
// (zeroed words, its status) -- otherwise its peers would wait in ncclAllGather / the host transport for ever -- and every
// rank reports the failure: the failing rank at once with its own message, the others when they collect the answer.
struct ShardState {
    int64_t n_words = 0;
    bool any_first = false;
    BufP ops;                    // int32 per word on the device
    BufP send[2], recv[2], merged[2], status[2];
    int64_t *status_host = nullptr;       // pinned: 2 slots x {status, rank}
    ~ShardState() { if (status_host) (void)hipHostFree(status_host); }
};

static CommState &comm_of(vdl_ctx *c) {
    if (!c->comm) throw Error(VDL_ERR_ARG, "no communicator on this context: call vdl_comm_init / vdl_comm_init_host first");
    return *c->comm;
}

// all ranks contribute `bytes` (a multiple of 8) from dev_send; dev_recv receives world * bytes, rank r's block at r * bytes
static void all_gather(vdl_ctx *c, const void *dev_send, void *dev_recv, size_t bytes, hipStream_t s) {
    CommState &m = comm_of(c);
    if (m.kind == CommState::RCCL) {
        RCCL_CHECK(rccl().AllGather(dev_send, dev_recv, bytes / 8, kNcclInt64, m.comm, s));
        return;
    }
    char *h = m.staging(bytes * (size_t)(m.world + 1));
    HIP_CHECK(hipMemcpyAsync(h, dev_send, bytes, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (m.host.all_gather(m.host.user, h, h + bytes, bytes)) throw Error(VDL_ERR_DEVICE, "the host transport's all_gather failed");
    HIP_CHECK(hipMemcpyAsync(dev_recv, h + bytes, bytes * (size_t)m.world, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));          // the staging buffer is reused by the next call
}

// `mine.size()` int64 words from every rank, rank after rank, on the host: the status / count exchanges of the routes below.  Every rank
// of a route must reach each of these exactly once per run, whatever happened to its local phase -- that is what they are for.
static std::vector<int64_t> gather_words(vdl_ctx *c, const std::vector<int64_t> &mine) {
    CommState &m = comm_of(c);
    if (m.world == 1) return mine;
    const size_t k = mine.size();
    std::vector<int64_t> all(k * (size_t)m.world);
    BufP dsend = dev_alloc(c, sizeof(int64_t) * k), drecv = dev_alloc(c, sizeof(int64_t) * k * (size_t)m.world);
    HIP_CHECK(hipMemcpyAsync(dsend->p, mine.data(), sizeof(int64_t) * k, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));                        // (`mine` is the caller's)
    all_gather(c, dsend->p, drecv->p, sizeof(int64_t) * k, c->stream);
    c->fetch_to_host(drecv->p, all.size(), all.data(), c->stream);
    return all;
}

// every column c of `ncols`: rows [soff[d], soff[d] + scnt[d]) of send + c * n_send go to rank d, which receives them at
// rows [roff[me-th source]..) of recv + c * n_recv, pieces in source-rank order
static void all_to_all_columns(vdl_ctx *c, const int64_t *dev_send, int64_t n_send, const std::vector<int64_t> &scnt, int64_t *dev_recv,
                               int64_t n_recv, const std::vector<int64_t> &rcnt, int ncols, hipStream_t s) {
    CommState &m = comm_of(c);
    std::vector<int64_t> soff((size_t)m.world + 1, 0), roff((size_t)m.world + 1, 0);
    for (int r = 0; r < m.world; r++) { soff[(size_t)r + 1] = soff[(size_t)r] + scnt[(size_t)r]; roff[(size_t)r + 1] = roff[(size_t)r] + rcnt[(size_t)r]; }
    if (m.kind == CommState::RCCL) {
        RCCL_CHECK(rccl().GroupStart());
        for (int r = 0; r < m.world; r++)
            for (int col = 0; col < ncols; col++) {
                if (scnt[(size_t)r] > 0)
                    RCCL_CHECK(rccl().Send(dev_send + (int64_t)col * n_send + soff[(size_t)r], (size_t)scnt[(size_t)r], kNcclInt64, r, m.comm, s));
                if (rcnt[(size_t)r] > 0)
                    RCCL_CHECK(rccl().Recv(dev_recv + (int64_t)col * n_recv + roff[(size_t)r], (size_t)rcnt[(size_t)r], kNcclInt64, r, m.comm, s));
            }
        RCCL_CHECK(rccl().GroupEnd());
        return;
    }
    // HOST transport: one all_to_all per call over a peer-major staging layout (block of peer r = its piece of every column)
    const size_t sb = sizeof(int64_t) * (size_t)ncols * (size_t)n_send, rb = sizeof(int64_t) * (size_t)ncols * (size_t)n_recv;
    char *h = m.staging(2 * (sb + rb) + 64);
    int64_t *hs = (int64_t *)h, *hsp = hs + (size_t)ncols * (size_t)n_send, *hrp = hsp + (size_t)ncols * (size_t)n_send, *hr = hrp + (size_t)ncols * (size_t)n_recv;
    if (sb) HIP_CHECK(hipMemcpyAsync(hs, dev_send, sb, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    std::vector<size_t> sbytes((size_t)m.world), rbytes((size_t)m.world);
    int64_t at = 0;
    for (int r = 0; r < m.world; r++) {
        for (int col = 0; col < ncols; col++) {
            std::memcpy(hsp + at, hs + (int64_t)col * n_send + soff[(size_t)r], sizeof(int64_t) * (size_t)scnt[(size_t)r]);
            at += scnt[(size_t)r];
        }
        sbytes[(size_t)r] = sizeof(int64_t) * (size_t)ncols * (size_t)scnt[(size_t)r];
        rbytes[(size_t)r] = sizeof(int64_t) * (size_t)ncols * (size_t)rcnt[(size_t)r];
    }
    if (m.host.all_to_all(m.host.user, hsp, sbytes.data(), hrp, rbytes.data())) throw Error(VDL_ERR_DEVICE, "the host transport's all_to_all failed");
    at = 0;
    for (int r = 0; r < m.world; r++)
        for (int col = 0; col < ncols; col++) {
            std::memcpy(hr + (int64_t)col * n_recv + roff[(size_t)r], hrp + at, sizeof(int64_t) * (size_t)rcnt[(size_t)r]);
            at += rcnt[(size_t)r];
        }
    if (rb) HIP_CHECK(hipMemcpyAsync(dev_recv, hr, rb, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
}

// the engine's launches go to `to` for the lifetime of this object (internal: no drain, the caller orders the streams with events)
struct StreamSwap {
    vdl_ctx *c;
    hipStream_t saved;
    StreamSwap(vdl_ctx *ctx, hipStream_t to) : c(ctx), saved(ctx->stream) { c->stream = to; }
    ~StreamSwap() { c->stream = saved; }
};

static ShardState &shard_state(vdl_ctx *c, vdl_plan *p, const CommState &m) {
    int64_t nw = 0;
    const int32_t *ops = nullptr;
    if (vdl_plan_partial_spec(p, &nw, &ops) != VDL_OK) throw Error(VDL_ERR_UNSUPPORTED, c->err);
    if (!p->shard) p->shard = std::make_shared<ShardState>();
    ShardState &st = *p->shard;
    const size_t words = (size_t)std::max<int64_t>(nw, 1);
    if (st.n_words != nw || !st.ops || !st.recv[0] || st.recv[0]->cls < sizeof(int64_t) * (2 * words + 1) * (size_t)m.world) {
        st.n_words = nw;
        st.any_first = false;
        for (int64_t i = 0; i < nw; i++) st.any_first |= ops[i] == VDL_REDUCE_FIRST;
        st.ops = dev_alloc(c, sizeof(int32_t) * words);
        if (nw) HIP_CHECK(hipMemcpyAsync(st.ops->p, ops, sizeof(int32_t) * (size_t)nw, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));                    // `ops` belongs to the plan and may be rewritten by the next spec call
        for (int k = 0; k < 2; k++) {
            st.send[k] = dev_alloc(c, sizeof(int64_t) * (2 * words + 1));
            st.recv[k] = dev_alloc(c, sizeof(int64_t) * (2 * words + 1) * (size_t)m.world);
            st.merged[k] = dev_alloc(c, sizeof(int64_t) * words);
            st.status[k] = dev_alloc(c, sizeof(int64_t) * 2);
        }
        if (!st.status_host) HIP_CHECK(hipHostMalloc((void **)&st.status_host, sizeof(int64_t) * 4, hipHostMallocDefault));
        for (int k = 0; k < 4; k++) st.status_host[k] = 0;
    }
    return st;
}

static void sharded_begin(vdl_ctx *c, vdl_plan *p, int slot) {
    need_device(c);
    if (slot < 0 || slot > 1) throw Error(VDL_ERR_ARG, "slot must be 0 or 1");
    CommState &m = comm_of(c);
    ShardState &st = shard_state(c, p, m);
    const int64_t nw = st.n_words;
    int64_t *send = (int64_t *)st.send[slot]->p;
    const bool overlap = m.kind == CommState::RCCL;
    hipStream_t cs = overlap ? m.stream : c->stream;
    if (overlap && m.ev_merged[slot]) HIP_CHECK(hipStreamWaitEvent(c->stream, m.ev_merged[slot], 0));    // the previous query of this slot has left the send buffer
    // the local phase; its outcome travels with the words (ShardState)
    int64_t local_rc = vdl_run_local(c, p, send);
    std::string local_error = local_rc != VDL_OK ? c->err : std::string();
    if (local_rc == VDL_OK) p->ev_buf[p->last_ev] = st.merged[slot]->p;            // the scan's timing is claimed by the finalisation of the MERGED words
    if (overlap) {
        if (!m.ev_local[slot]) {
            HIP_CHECK(hipEventCreateWithFlags(&m.ev_local[slot], hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&m.ev_merged[slot], hipEventDisableTiming));
        }
        HIP_CHECK(hipEventRecord(m.ev_local[slot], c->stream));
        HIP_CHECK(hipStreamWaitEvent(cs, m.ev_local[slot], 0));
    }
    StreamSwap on_comm(c, cs);
    // second half of the send buffer: the words again, FoldChoose words resolved to the value at this rank's own row id
    if (local_rc == VDL_OK) {
        if (nw) HIP_CHECK(hipMemcpyAsync(send + nw, send, sizeof(int64_t) * (size_t)nw, hipMemcpyDeviceToDevice, cs));
        if (st.any_first && vdl_resolve_first(c, p, send + nw) != VDL_OK) { local_rc = VDL_ERR_DEVICE; local_error = c->err; }
    }
    if (local_rc != VDL_OK && nw) HIP_CHECK(hipMemsetAsync(send, 0, sizeof(int64_t) * 2 * (size_t)nw, cs));
    int64_t *slot_status = st.status_host + 2 * slot;
    slot_status[0] = local_rc;                 // (pinned: read by the copy below when the stream gets there; this slot's previous
    slot_status[1] = m.rank;                   //  query has been collected -- _end -- before the slot is begun again)
    HIP_CHECK(hipMemcpyAsync(send + 2 * nw, slot_status, sizeof(int64_t), hipMemcpyHostToDevice, cs));
    int64_t *merged = (int64_t *)st.merged[slot]->p;
    const int64_t stride = 2 * nw + 1;
    if (m.world > 1) {
        all_gather(c, send, st.recv[slot]->p, sizeof(int64_t) * (size_t)stride, cs);
        HIP_CHECK(launch_merge_words((const int64_t *)st.recv[slot]->p, m.world, nw, stride, (const int32_t *)st.ops->p, merged, (int64_t *)st.status[slot]->p, cs));
    } else {
        HIP_CHECK(launch_merge_words(send, 1, nw, stride, (const int32_t *)st.ops->p, merged, (int64_t *)st.status[slot]->p, cs));
    }
    if (overlap) HIP_CHECK(hipEventRecord(m.ev_merged[slot], cs));
    if (local_rc != VDL_OK) {
        // this rank has done its part of the collective; it reports its own failure now
        HIP_CHECK(hipStreamSynchronize(cs));
        throw Error((int)local_rc, local_error);
    }
    HIP_CHECK(hipMemcpyAsync(slot_status, st.status[slot]->p, sizeof(int64_t) * 2, hipMemcpyDeviceToHost, cs));   // ahead of the words: same event
    if (vdl_finalize_begin(c, p, merged, slot) != VDL_OK) throw Error(VDL_ERR_DEVICE, c->err);
    if (!(p->use_fusion && p->fused.ok)) HIP_CHECK(hipStreamSynchronize(cs));     // (fold records: finalised at once, no slot event for _end to wait on)
}

// after vdl_finalize_end has waited for the slot: did every rank's local phase succeed?
static void sharded_check_status(vdl_plan *p, int slot) {
    if (!p->shard || !p->shard->status_host || slot < 0 || slot > 1) return;
    const int64_t *s = p->shard->status_host + 2 * slot;
    if (s[0] != VDL_OK)
        throw Error(VDL_ERR_UNSUPPORTED, "sharded run: the local phase failed on rank " + std::to_string(s[1]) + " (status " + std::to_string(s[0]) +
                                         "); the merged words are not an answer");
}

static void sharded_exchange(vdl_ctx *c, vdl_plan *p) {
    need_device(c);
    CommState &m = comm_of(c);
    int ncols = 0;
    const std::string table = p->sharded_table;
    if (table.empty()) throw Error(VDL_ERR_ARG, "vdl_run_sharded: name the row-sharded table first (vdl_plan_set_sharded_table)");
    struct AllowFolds { vdl_plan *p; AllowFolds(vdl_plan *q) : p(q) { p->ex_allow_folds = true; } ~AllowFolds() { p->ex_allow_folds = false; } } allow(p);
    if (vdl_exchange_spec(p, table.c_str(), &ncols) != VDL_OK) throw Error(VDL_ERR_UNSUPPORTED, c->err);
    // local phase; its outcome travels with the counts so that no rank is left waiting in a collective after a failure elsewhere.
    // Global folds over the sharded table that the tail reads beside the Partition (Q11's HAVING threshold) travel in the same
    // all-gather: three mergeable words each, merged on the host below.
    const size_t n_fold_words = 3 * exchange_fold_count(p, table);
    // Round 4: WHERE the key domain is cut follows the data.  Every rank counts its keys in kExBins equal slices of the pivots' domain;
    // the histograms travel in the one all-gather (with the status and the fold words); every rank sums them and cuts the slices into
    // `world` runs of about equal population -- the same cut everywhere --, routes its rows by it, and reads what it will receive from
    // whom out of the gathered histograms.  (With the DECLARED domain cut evenly the last three of eight ranks received nothing for
    // TPC-H Q3: the order keys in use reach 0.56 of their power-of-two domain.)  Owners stay contiguous key ranges in rank order, so the
    // ranks' outputs still concatenate to the unsharded result.
    const size_t row = 1 + (size_t)kExBins + 1 + n_fold_words + 1;      // {status, histogram, keys outside the pivots, fold words, holes}
    std::vector<int64_t> mine(row, 0);
    std::string local_error;
    int rc = guard(c, [&] {
        exchange_local(c, p, m.world);
        exchange_histogram(c, p, mine.data() + 1);
    });
    if (rc == VDL_OK && mine[1 + (size_t)kExBins] > 0) {
        rc = VDL_ERR_UNSUPPORTED;
        c->err = std::to_string(mine[1 + (size_t)kExBins]) + " row(s) carry a partition key outside the pivots; run unsharded";
    }
    if (rc != VDL_OK) { local_error = c->err; mine.assign(row, 0); }
    else if (p->ex.fold_words.size() == n_fold_words) std::copy(p->ex.fold_words.begin(), p->ex.fold_words.end(), mine.begin() + 2 + kExBins);
    mine[0] = rc;
    // (does any travelling vector of this rank have empty slots of its own?  If nobody's has, the mask column that carries the vectors'
    // validity is neither written nor sent nor read: one column in four of TPC-H Q18's exchange)
    mine[row - 1] = rc == VDL_OK && exchange_has_holes(p) ? 1 : 0;
    const std::vector<int64_t> all = gather_words(c, mine);
    bool holes = false;
    for (int r = 0; r < m.world; r++) {
        if (all[(size_t)r * row] != VDL_OK) {
            if (rc != VDL_OK) throw Error(rc, local_error);
            throw Error(VDL_ERR_UNSUPPORTED, "sharded Partition exchange failed on rank " + std::to_string(r));
        }
        holes |= all[(size_t)r * row + row - 1] != 0;
    }
    p->ex.skip_mask = !holes;
    const int sent_cols = holes ? ncols : ncols - 1;
    if (n_fold_words) {
        // merge: value by the fold's reduction, first row by MIN, count by SUM (vdl_exchange.cpp: k_fold_words / k_fold_record)
        std::vector<int64_t> merged(n_fold_words);
        for (size_t w = 0; w < n_fold_words; w++) {
            const int kind = w % 3 == 0 ? exchange_fold_kind(p, w / 3) : (w % 3 == 1 ? 1 : 0);      // 0 sum, 1 min, 2 max
            int64_t acc = all[2 + (size_t)kExBins + w];
            for (int r = 1; r < m.world; r++) {
                const int64_t v = all[(size_t)r * row + 2 + (size_t)kExBins + w];
                acc = kind == 1 ? std::min(acc, v) : kind == 2 ? std::max(acc, v) : (int64_t)((uint64_t)acc + (uint64_t)v);
            }
            merged[w] = acc;
        }
        p->ex.fold_merged = merged;
    }
    // the cut: slice b goes to the rank in whose share of the population its middle lies (monotone in b)
    std::vector<int64_t> global((size_t)kExBins, 0);
    int64_t total = 0;
    for (int r = 0; r < m.world; r++)
        for (int b = 0; b < kExBins; b++) { global[(size_t)b] += all[(size_t)r * row + 1 + (size_t)b]; }
    for (int b = 0; b < kExBins; b++) total += global[(size_t)b];
    std::vector<int32_t> owner((size_t)kExBins, 0);
    {
        int64_t before = 0;
        for (int b = 0; b < kExBins; b++) {
            const int64_t mid = before + global[(size_t)b] / 2;
            int64_t o = total > 0 ? (int64_t)(((unsigned __int128)(uint64_t)mid * (uint64_t)m.world) / (uint64_t)total) : ((int64_t)b * m.world) / kExBins;
            owner[(size_t)b] = (int32_t)std::min<int64_t>(o, m.world - 1);
            before += global[(size_t)b];
        }
    }
    std::vector<int64_t> scnt((size_t)m.world, 0), rcnt((size_t)m.world, 0);
    if (guard(c, [&] { exchange_route(c, p, owner.data(), scnt.data()); }) != VDL_OK) throw Error(VDL_ERR_DEVICE, c->err);
    int64_t n_send = 0, n_recv = 0;
    for (int r = 0; r < m.world; r++) {
        for (int b = 0; b < kExBins; b++) if (owner[(size_t)b] == m.rank) rcnt[(size_t)r] += all[(size_t)r * row + 1 + (size_t)b];
        n_send += scnt[(size_t)r]; n_recv += rcnt[(size_t)r];
    }
    BufP send = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(n_send * ncols, 1));
    if (vdl_exchange_pack(c, p, send->p) != VDL_OK) throw Error(VDL_ERR_DEVICE, c->err);
    BufP recv = send;
    if (m.world > 1) {
        recv = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(n_recv * ncols, 1));
        all_to_all_columns(c, (const int64_t *)send->p, n_send, scnt, (int64_t *)recv->p, n_recv, rcnt, sent_cols, c->stream);
    }
    p->shard_keep = recv;                                     // the tail reads the received columns in place
    if (vdl_exchange_finish(c, p, recv->p, n_recv) != VDL_OK) throw Error(VDL_ERR_DEVICE, c->err);
    p->shard_keep.reset();
}

// A fused plan with a semi-join set (EXISTS / IN with the dimension on the left: TPC-H Q4, Vlite.hs:1212-1222) when the set's SOURCE
// table is the row-sharded one and every table the scans read is replicated: each rank builds the set from its rows, ONE all-gather
// moves the sets (150 M bits = 19 MB per rank at SF100), a kernel ORs them and clips the union at the source table's global
// length, and every rank then runs the scans of the replicated tables against the complete set: every rank ends with the whole
// answer.  "" = this plan and placement qualify, else why not.
static std::string semi_route_refusal(const vdl_plan *p) {
    if (!(p->use_fusion && p->fused.ok)) return "the plan is not fused";
    const std::string &t = p->sharded_table;
    if (t.empty()) return "name the row-sharded table first (vdl_plan_set_sharded_table)";
    bool any = false;
    for (const PreludeItem &it : p->fused.prelude) {
        if (it.kind == PreludeItem::SEMI_BITMAP) {
            if (it.table != t) return "the semi-join set is built from table '" + it.table + "', which is not the sharded one";
            any = true;
        } else if (it.kind == PreludeItem::DIM_BITMAP && it.table == t) return "a dimension selection is taken over the sharded table";
    }
    if (!any) return "the plan builds no semi-join set";
    for (const ScanPlan &sp : p->fused.scans) if (sp.table == t) return "a scan reads the sharded table itself";
    for (const GroupScanPlan &gp : p->fused.gscans) if (gp.table == t) return "a scan reads the sharded table itself";
    return "";
}
// Every rank exchanges {status, which sets it holds, their sizes} BEFORE the sets travel, and does so exactly once per run: from the
// hook behind the prelude when that ran, otherwise -- the prelude threw (a missing column, a HIP error), or the fused plan was
// abandoned for this data and the hook never came -- from here, with its failure, so that no peer is left waiting in ncclAllGather.
static void sharded_semi(vdl_ctx *c, vdl_plan *p) {
    need_device(c);
    CommState &m = comm_of(c);
    struct Restore { vdl_plan *p; ~Restore() { p->after_prelude = nullptr; p->semi_unclamped = false; } } restore{p};
    p->semi_unclamped = true;
    bool met = false;                                          // this rank has taken part in the status exchange of this run
    auto meet = [&m, &met](vdl_ctx *cc, int64_t status, int64_t mask, int64_t words) {
        met = true;
        const std::vector<int64_t> all = gather_words(cc, {status, mask, words});
        for (int r = 0; r < m.world; r++)
            if (all[(size_t)r * 3] != VDL_OK) {
                if (status != VDL_OK) return;                  // (this rank reports its own error)
                throw Error(VDL_ERR_UNSUPPORTED, "sharded run: the semi-join sets could not be built on rank " + std::to_string(r) + " (its error is reported there)");
            }
        for (int r = 0; r < m.world; r++)
            if (all[(size_t)r * 3 + 1] != mask || all[(size_t)r * 3 + 2] != words)
                throw Error(VDL_ERR_SHAPE, "sharded run: rank " + std::to_string(r) + " builds other semi-join sets than rank " + std::to_string(m.rank) + " (the replicated tables differ)");
    };
    p->after_prelude = [&m, &meet](vdl_ctx *cc, vdl_plan *pp) {
        const FusedPlan &F = pp->fused;
        int64_t mask = 0, total_words = 0;
        for (size_t k = 0; k < F.prelude.size(); k++)
            if (F.prelude[k].kind == PreludeItem::SEMI_BITMAP && pp->prelude_buf[k]) {
                mask |= (int64_t)1 << (k & 62);
                total_words += std::max<int64_t>((pp->prelude_n[k] + 63) >> 6, 1);
            }
        meet(cc, VDL_OK, mask, total_words);
        for (size_t k = 0; k < F.prelude.size(); k++) {
            if (F.prelude[k].kind != PreludeItem::SEMI_BITMAP || !pp->prelude_buf[k]) continue;
            const int64_t words = std::max<int64_t>((pp->prelude_n[k] + 63) >> 6, 1);
            BufP send = dev_alloc(cc, sizeof(uint64_t) * (size_t)(words + 1)), recv = dev_alloc(cc, sizeof(uint64_t) * (size_t)(words + 1) * (size_t)m.world);
            HIP_CHECK(hipMemcpyAsync(send->p, pp->prelude_buf[k]->p, sizeof(uint64_t) * (size_t)words, hipMemcpyDeviceToDevice, cc->stream));
            const int64_t rows = pp->prelude_rows[k];
            HIP_CHECK(hipMemcpyAsync((uint64_t *)send->p + words, &rows, sizeof rows, hipMemcpyHostToDevice, cc->stream));
            HIP_CHECK(hipStreamSynchronize(cc->stream));                          // (`rows` is a local)
            all_gather(cc, send->p, recv->p, sizeof(uint64_t) * (size_t)(words + 1), cc->stream);
            HIP_CHECK(launch_or_sets((const uint64_t *)recv->p, m.world, words, (uint64_t *)pp->prelude_buf[k]->p, cc->stream));
        }
    };
    const int rc = vdl_run(c, p);
    const std::string own = c->err;
    bool abandoned = false;
    std::string label;
    for (const Timing &t : p->timings)
        if (t.label.find("fusedPlanAbandoned") != std::string::npos) { abandoned = true; label = t.label; }
    if (!met) meet(c, rc != VDL_OK ? rc : (int64_t)VDL_ERR_UNSUPPORTED, 0, 0);      // the hook never ran: say so to the peers, who stop with this rank
    if (rc != VDL_OK) throw Error(rc, own);
    if (abandoned)
        throw Error(VDL_ERR_UNSUPPORTED, "sharded run: the fused plan does not hold for this data (" + label + "), and statement by statement the plan has no sharded route");
}

// ---- the "front" route: a plan whose work on the sharded table is a fused front (select + take: vdl_fuse.h ProjPlan) ----
// Every rank runs the front over its rows; the survivors' vectors (a few per cent of the table for the TPC-H plans) are
// all-gathered, rank after rank = global row order, and every rank runs the statements above the front on the complete vectors:
// any number of Partitions, folds over folds, lookups in the replicated tables -- whatever the program says, since from there on
// it is the unsharded program on the unsharded data.  Every rank ends with the whole answer.  It scales the scan of the sharded
// table (the part that grows with it), not the tail; plans whose tail is heavy and has ONE Partition take the exchange route,
// which is tried first.  TPC-H Q16 (two Partitions: count(distinct ..) under a GROUP BY) runs this way.
// "" = the plan and placement qualify, else why not.
static std::string front_route_refusal(const vdl_plan *p) {
    const ProjPlan &J = p->fused.proj;
    const std::string &t = p->sharded_table;
    if (t.empty()) return "name the row-sharded table first (vdl_plan_set_sharded_table)";
    if (!p->use_fusion || (p->fused.ok)) return "the plan has no fused front (it fuses as a whole, or fusion is off)";
    if (!J.ok) return "the plan has no fused front: " + J.why;
    if (J.table != t) return "the fused front scans table '" + J.table + "', not the sharded one";
    for (const PreludeItem &it : p->fused.prelude)
        if (it.table == t) return "a dimension-side scan of the front reads the sharded table";
    // nothing above the front may read the sharded table by itself (its length included): walk down from the outputs, stopping
    // at the statements the front produces
    const Program &P = p->prog;
    std::vector<char> seen(P.nodes.size(), 0);
    std::vector<int> stack(P.outputs.begin(), P.outputs.end());
    for (int id : J.nodes) if (id > 0 && (size_t)id < seen.size()) seen[(size_t)id] = 2;
    while (!stack.empty()) {
        const int id = stack.back(); stack.pop_back();
        if (id <= 0 || (size_t)id >= seen.size() || seen[(size_t)id]) continue;
        seen[(size_t)id] = 1;
        const Node &n = P.at(id);
        if (n.op == Op::Load && n.column.compare(0, t.size() + 1, t + ".") == 0)
            return "statement " + std::to_string(id) + " above the front reads " + n.column + " of the sharded table";
        for (int o : {n.a, n.b, n.c}) if (o > 0) stack.push_back(o);
    }
    return "";
}

// the same `ncols` columns from every rank, rank after rank: column k of rank r (cnt[r] int64) lands at recv[k] + sum(cnt[0..r))
static void all_gather_rows(vdl_ctx *c, const std::vector<const int64_t *> &send, const std::vector<int64_t *> &recv, const std::vector<int64_t> &cnt, hipStream_t s) {
    CommState &m = comm_of(c);
    const int ncols = (int)send.size();
    std::vector<int64_t> off((size_t)m.world + 1, 0);
    int64_t biggest = 0;
    for (int r = 0; r < m.world; r++) { off[(size_t)r + 1] = off[(size_t)r] + cnt[(size_t)r]; biggest = std::max(biggest, cnt[(size_t)r]); }
    const int64_t mine = cnt[(size_t)m.rank];
    if (m.kind == CommState::RCCL) {
        RCCL_CHECK(rccl().GroupStart());
        for (int k = 0; k < ncols; k++)
            for (int r = 0; r < m.world; r++) {
                if (mine > 0) RCCL_CHECK(rccl().Send(send[(size_t)k], (size_t)mine, kNcclInt64, r, m.comm, s));
                if (cnt[(size_t)r] > 0) RCCL_CHECK(rccl().Recv(recv[(size_t)k] + off[(size_t)r], (size_t)cnt[(size_t)r], kNcclInt64, r, m.comm, s));
            }
        RCCL_CHECK(rccl().GroupEnd());
        return;
    }
    // HOST transport: one all_gather of blocks padded to the largest rank's rows (a rehearsal transport: simplicity over bytes)
    if (biggest == 0) return;
    const size_t block = sizeof(int64_t) * (size_t)ncols * (size_t)biggest;
    char *h = m.staging(block * (size_t)(m.world + 1));
    int64_t *hs = (int64_t *)h, *hr = (int64_t *)(h + block);
    for (int k = 0; k < ncols; k++)
        if (mine > 0) HIP_CHECK(hipMemcpyAsync(hs + (size_t)k * (size_t)biggest, send[(size_t)k], sizeof(int64_t) * (size_t)mine, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (m.host.all_gather(m.host.user, hs, hr, block)) throw Error(VDL_ERR_DEVICE, "the host transport's all_gather failed");
    for (int r = 0; r < m.world; r++)
        for (int k = 0; k < ncols; k++)
            if (cnt[(size_t)r] > 0)
                HIP_CHECK(hipMemcpyAsync(recv[(size_t)k] + off[(size_t)r], hr + ((size_t)r * (size_t)ncols + (size_t)k) * (size_t)biggest, sizeof(int64_t) * (size_t)cnt[(size_t)r],
                                         hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));          // the staging buffer is reused by the next call
}

static void sharded_front(vdl_ctx *c, vdl_plan *p) {
    need_device(c);
    CommState &m = comm_of(c);
    const ProjPlan &J = p->fused.proj;
    // this rank's rows of the table and everybody's: the front's row-id columns count from the table's first row
    int64_t n_local = -1;
    for (const auto &kv : c->cols)
        if (kv.first.compare(0, J.table.size() + 1, J.table + ".") == 0 && kv.first.find(".heap") == std::string::npos) { n_local = kv.second.n; break; }
    // {status, rows}: a rank without the table says so here and everybody stops together
    const std::vector<int64_t> rows = gather_words(c, {n_local < 0 ? (int64_t)VDL_ERR_ARG : (int64_t)VDL_OK, std::max<int64_t>(n_local, 0)});
    int64_t row0 = 0, n_global = 0;
    for (int r = 0; r < m.world; r++) {
        if (rows[(size_t)r * 2] != VDL_OK) throw Error(VDL_ERR_ARG, "sharded run: rank " + std::to_string(r) + " holds no column of table '" + J.table + "'");
        if (r < m.rank) row0 += rows[(size_t)r * 2 + 1];
        n_global += rows[(size_t)r * 2 + 1];
    }
    struct Restore { vdl_plan *p; int64_t row_offset; ~Restore() { p->after_front = nullptr; p->row_offset = row_offset; p->front_rowid_global = false; } } restore{p, p->row_offset};
    p->row_offset = row0;
    p->front_rowid_global = true;
    p->after_front = [&m, row0, n_global](vdl_ctx *cc, vdl_plan *pp, std::map<int, DVec> &over, bool front, const std::string &failure) {
        const ProjPlan &JJ = pp->fused.proj;
        hipStream_t s = cc->stream;
        // what travels: the survivors' row ids (made global) and every distinct packed vector of the front
        SelP sel;
        std::vector<BufP> bufs;                                 // [0] = row ids
        if (front) {
            for (const auto &kv : over) if (kv.second.kind == DVec::SPARSE && kv.second.sel) { sel = kv.second.sel; break; }
            if (sel) {
                bufs.push_back(sel->idx);
                for (int id : JJ.nodes) {
                    const auto at = over.find(id);             // (a missing vector is a failure to REPORT below, not to throw ahead of the collective)
                    if (at == over.end()) { sel.reset(); break; }
                    const DVec &v = at->second;
                    if (v.kind != DVec::SPARSE || v.sel != sel || v.valid) { sel.reset(); break; }
                    if (std::find(bufs.begin(), bufs.end(), v.data) == bufs.end()) bufs.push_back(v.data);
                }
            }
        }
        const bool ok = front && sel;
        const std::vector<int64_t> all = gather_words(cc, {ok ? (int64_t)VDL_OK : (int64_t)VDL_ERR_UNSUPPORTED, ok ? sel->m : 0});
        if (!ok) throw Error(VDL_ERR_UNSUPPORTED, "sharded run: the fused front did not run on this rank (" + (failure.empty() ? (pp->fallback_note.empty() ? std::string("its vectors do not share one selection") : pp->fallback_note) : failure) +
                                                   "), and statement by statement the plan has no sharded route");
        std::vector<int64_t> cnt((size_t)m.world);
        int64_t total = 0;
        for (int r = 0; r < m.world; r++) {
            if (all[(size_t)r * 2] != VDL_OK) throw Error(VDL_ERR_DEVICE, "sharded run: the fused front failed on rank " + std::to_string(r) + " (its error is reported there)");
            cnt[(size_t)r] = all[(size_t)r * 2 + 1];
            total += cnt[(size_t)r];
        }
        // global row ids of this rank's survivors
        BufP gids = dev_alloc(cc, sizeof(int64_t) * (size_t)std::max<int64_t>(sel->m, 1));
        if (sel->m > 0) {
            Src a; a.p = sel->idx->p; a.kind = SRC_I64;
            Src b; b.kind = SRC_RANGE; b.from = row0; b.step = 0;
            HIP_CHECK(launch_binary(B_ADD, a, b, (int64_t *)gids->p, sel->m, s));
        }
        std::vector<const int64_t *> send;
        std::vector<int64_t *> recv;
        std::vector<BufP> got;
        for (size_t k = 0; k < bufs.size(); k++) {
            send.push_back(k == 0 ? (const int64_t *)gids->p : (const int64_t *)bufs[k]->p);
            got.push_back(dev_alloc(cc, sizeof(int64_t) * (size_t)std::max<int64_t>(total, 1)));
            recv.push_back((int64_t *)got.back()->p);
        }
        all_gather_rows(cc, send, recv, cnt, s);
        auto whole = std::make_shared<Sel>();
        whole->n = n_global; whole->m = total; whole->idx = got[0]; whole->first_slot = -1;
        for (int id : JJ.nodes) {
            DVec &v = over.at(id);
            const size_t k = (size_t)(std::find(bufs.begin(), bufs.end(), v.data) - bufs.begin());
            v.n = n_global; v.sel = whole; v.data = got[k];
        }
    };
    if (vdl_run(c, p) != VDL_OK) throw Error(VDL_ERR_DEVICE, c->err);
}

// ---- the "replicate" route: the last resort, for a plan no other route serves (TPC-H Q18: it groups ALL lineitems by order before it
// filters anything, feeds a semi-join set from that and scans lineitem a second time) ----
// The columns of the sharded table that the program loads are all-gathered ONCE per catalog state into plan-owned buffers (rank after
// rank = row order), and every rank runs the whole program over them: no scaling of the query itself, but a correct answer on every
// rank from the placement the other plans of the session use, and no communication at all from the second run on.
// every rank's block of `bytes[r]` bytes, rank after rank, into recv
static void all_gather_bytes(vdl_ctx *c, const void *send, void *recv, const std::vector<int64_t> &bytes, hipStream_t s) {
    CommState &m = comm_of(c);
    std::vector<int64_t> off((size_t)m.world + 1, 0);
    int64_t biggest = 0;
    for (int r = 0; r < m.world; r++) { off[(size_t)r + 1] = off[(size_t)r] + bytes[(size_t)r]; biggest = std::max(biggest, bytes[(size_t)r]); }
    const int64_t mine = bytes[(size_t)m.rank];
    if (m.kind == CommState::RCCL) {
        RCCL_CHECK(rccl().GroupStart());
        for (int r = 0; r < m.world; r++) {
            if (mine > 0) RCCL_CHECK(rccl().Send(send, (size_t)mine, kNcclInt8, r, m.comm, s));
            if (bytes[(size_t)r] > 0) RCCL_CHECK(rccl().Recv((char *)recv + off[(size_t)r], (size_t)bytes[(size_t)r], kNcclInt8, r, m.comm, s));
        }
        RCCL_CHECK(rccl().GroupEnd());
        return;
    }
    if (biggest == 0) return;
    const size_t block = ((size_t)biggest + 7) & ~(size_t)7;
    char *h = m.staging(block * (size_t)(m.world + 1));
    if (mine > 0) HIP_CHECK(hipMemcpyAsync(h, send, (size_t)mine, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (m.host.all_gather(m.host.user, h, h + block, block)) throw Error(VDL_ERR_DEVICE, "the host transport's all_gather failed");
    for (int r = 0; r < m.world; r++)
        if (bytes[(size_t)r] > 0) HIP_CHECK(hipMemcpyAsync((char *)recv + off[(size_t)r], h + block * (size_t)(r + 1), (size_t)bytes[(size_t)r], hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
}

static void sharded_replicate(vdl_ctx *c, vdl_plan *p) {
    need_device(c);
    CommState &m = comm_of(c);
    const std::string &t = p->sharded_table;
    // Whether the table has to be gathered (again) is decided by ALL ranks on EVERY run -- one exchange of {needs it, status, rows, row
    // bytes, bytes this device can spare}: a rank whose catalog alone moved (a column uploaded again, a retry after an error) would
    // otherwise enter the collectives below by itself and wait there for ever.  The same words carry what the replica will cost, so
    // that a table that does not fit is refused by every rank together, with the reason, instead of an out-of-memory on some.
    std::vector<std::string> names;
    for (int id : p->prog.order) {
        const Node &n = p->prog.at(id);
        if (n.op != Op::Load || n.column.compare(0, t.size() + 1, t + ".") != 0 || n.column.find(".heap") != std::string::npos) continue;
        if (std::find(names.begin(), names.end(), n.column) == names.end()) names.push_back(n.column);
    }
    int64_t n_local = -1, status = VDL_OK, row_bytes = 0;
    for (const std::string &name : names) {                     // every rank holds the same columns, all as long as each other
        auto it = c->cols.find(name);
        if (it == c->cols.end()) { status = VDL_ERR_COLUMN; break; }
        if (n_local >= 0 && it->second.n != n_local) { status = VDL_ERR_SHAPE; break; }
        n_local = it->second.n;
        row_bytes += (int64_t)it->second.width;
    }
    const bool stale = p->replica.empty() || p->replica_version != c->catalog_version;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    int64_t held = 0;                                           // what a stale replica of this plan gives back before the new one is made
    for (const auto &kv : p->replica) held += kv.second.n * (int64_t)kv.second.width;
    const std::vector<int64_t> all = gather_words(c, {stale ? 1 : 0, status, std::max<int64_t>(n_local, 0), row_bytes, (int64_t)free_b + (int64_t)c->pool->idle_bytes() + held});
    bool gather = false;
    int64_t n_global = 0;
    for (int r = 0; r < m.world; r++) {
        gather |= all[(size_t)r * 5] != 0;
        if (all[(size_t)r * 5 + 1] != VDL_OK || all[(size_t)r * 5 + 3] != row_bytes)
            throw Error(VDL_ERR_COLUMN, "sharded run: rank " + std::to_string(r) + " does not hold the columns of table '" + t + "' this plan loads (or they differ in length or width)");
        n_global += all[(size_t)r * 5 + 2];
    }
    if (gather) {
        const int64_t need = n_global * row_bytes;
        for (int r = 0; r < m.world; r++)
            if (need + (need >> 3) > all[(size_t)r * 5 + 4])    // (an eighth of headroom for the run itself)
                throw Error(VDL_ERR_DEVICE, "sharded run: the replicate route needs " + std::to_string(need) + " bytes per rank for table '" + t + "' (" + std::to_string(n_global) +
                                            " rows x " + std::to_string(row_bytes) + " B of loaded columns); rank " + std::to_string(r) + " can spare " + std::to_string(all[(size_t)r * 5 + 4]));
        p->replica.clear();
        for (const std::string &name : names) {
            const Column &mine = c->cols.at(name);
            std::vector<int64_t> bytes((size_t)m.world);
            for (int r = 0; r < m.world; r++) bytes[(size_t)r] = all[(size_t)r * 5 + 2] * (int64_t)mine.width;
            Column whole;
            whole.width = mine.width; whole.n = n_global;
            whole.owned = dev_alloc(c, (size_t)std::max<int64_t>(n_global * (int64_t)mine.width, 8));
            whole.dev = whole.owned->p;
            all_gather_bytes(c, mine.dev, whole.owned->p, bytes, c->stream);
            p->replica[name] = whole;
        }
        HIP_CHECK(hipStreamSynchronize(c->stream));
        p->replica_version = c->catalog_version;
    }
    // (the overlay changes what find_col hands out under an unchanged catalog: bindings and kernels specialised for column addresses are
    // keyed by binding_version(), which moves with it)
    struct Restore { vdl_ctx *c; vdl_plan *p; int64_t row_offset; ~Restore() { c->overlay = nullptr; c->overlay_epoch++; p->row_offset = row_offset; } } restore{c, p, p->row_offset};
    c->overlay = &p->replica;
    c->overlay_epoch++;
    p->row_offset = 0;                                          // every rank runs the whole table
    if (vdl_run(c, p) != VDL_OK) throw Error(VDL_ERR_DEVICE, c->err);
}

// ---- the "chain" route (TPC-H Q18; analysis and the reasoning: vdl_exchange.cpp analyse_chain) ----
// stage 1: the exchange route's own run over the rewritten program with the operands of the position sets for outputs (every rank ends
// with the packed positions its key range contributes); merge: one all-gather of {status, positions and length per set}, one grouped
// send / receive per set, every rank builds the same sets; stage 2: the rest with the sets given -- per-row work on each rank's own
// shard, the rows that reach the next Partition all-gathered (rank after rank = row order), the tail on every rank.  Every rank reaches
// every collective once per run whatever fails where: a failure travels in the next status exchange.
static bool chain_route(vdl_plan *p, std::string *why = nullptr) {
    if (getenv("VDL_NO_CHAIN_ROUTE")) { if (why) *why = "switched off (VDL_NO_CHAIN_ROUTE)"; return false; }
    std::string w;
    const bool ok = chain_plan(p, w) != nullptr;
    if (why) *why = w;
    return ok;
}

static void sharded_chain(vdl_ctx *c, vdl_plan *p) {
    need_device(c);
    CommState &m = comm_of(c);
    std::string why;
    const std::shared_ptr<ChainPlan> cp = chain_plan(p, why);
    if (!cp) throw Error(VDL_ERR_UNSUPPORTED, "no chain route: " + why);
    struct Restore {
        vdl_plan *p; Program saved;
        ~Restore() { p->prog = std::move(saved); p->chain.stage = 0; p->chain.lists.clear(); p->chain.sets.clear(); p->shard_keep.reset(); }
    } restore{p, p->prog};
    const std::vector<int> outputs = p->prog.outputs;
    p->prog = cp->prog;
    const size_t nsets = cp->sets.size();
    // ---- stage 1
    p->chain.stage = 1;
    p->prog.outputs = cp->targets;
    std::sort(p->prog.outputs.begin(), p->prog.outputs.end());
    p->prog.outputs.erase(std::unique(p->prog.outputs.begin(), p->prog.outputs.end()), p->prog.outputs.end());
    int rc = VDL_OK;
    std::string own;
    try { sharded_exchange(c, p); }
    catch (const Error &e) { rc = e.code; own = e.what(); }
    if (rc == VDL_OK && p->chain.lists.size() != nsets) { rc = VDL_ERR_DEVICE; own = "the position sets were not collected"; }
    p->prog.outputs = outputs;
    // ---- merge
    std::vector<int64_t> mine(1 + 2 * nsets, 0);
    mine[0] = rc;
    if (rc == VDL_OK) for (size_t k = 0; k < nsets; k++) { mine[1 + 2 * k] = p->chain.lists[k].m; mine[2 + 2 * k] = p->chain.lists[k].len; }
    std::vector<int64_t> all = gather_words(c, mine);
    const size_t row = mine.size();
    for (int r = 0; r < m.world; r++)
        if (all[(size_t)r * row] != VDL_OK) {
            if (rc != VDL_OK) throw Error(rc, own);
            throw Error(VDL_ERR_UNSUPPORTED, "sharded run: the GROUP BY behind the position sets failed on rank " + std::to_string(r) + " (its error is reported there)");
        }
    for (size_t k = 0; k < nsets; k++) {
        std::vector<int64_t> cnt((size_t)m.world);
        int64_t total = 0, len = 0;
        for (int r = 0; r < m.world; r++) {
            cnt[(size_t)r] = all[(size_t)r * row + 1 + 2 * k];
            total += cnt[(size_t)r];
            const int64_t l = all[(size_t)r * row + 2 + 2 * k];
            len = cp->size_replicated[k] ? std::max(len, l) : len + l;
        }
        BufP got = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(total, 1));
        all_gather_rows(c, {(const int64_t *)p->chain.lists[k].list->p}, {(int64_t *)got->p}, cnt, c->stream);
        chain_build_set(c, p, k, got, total, len);
        HIP_CHECK(hipStreamSynchronize(c->stream));          // (`got` goes)
    }
    p->chain.lists.clear();
    // ---- stage 2
    p->chain.stage = 2;
    if (!cp->second_cut) {
        // nothing above the sets reads the sharded table: the rest is the same on every rank
        if (guard(c, [&] { chain_run_everywhere(c, p); }) != VDL_OK) throw Error(VDL_ERR_DEVICE, c->err);
        return;
    }
    int ncols = 0;
    int64_t n_send = 0;
    BufP send;
    rc = guard(c, [&] {
        if (vdl_exchange_spec(p, p->sharded_table.c_str(), &ncols) != VDL_OK) throw Error(VDL_ERR_UNSUPPORTED, c->err);
        exchange_local(c, p, 1);                              // "one destination": every row with a key, in row order
        exchange_route(c, p, nullptr, &n_send);
        send = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(n_send * ncols, 1));
        if (vdl_exchange_pack(c, p, send->p) != VDL_OK) throw Error(VDL_ERR_DEVICE, c->err);
    });
    own = c->err;
    all = gather_words(c, {rc, rc == VDL_OK ? n_send : 0});
    std::vector<int64_t> cnt((size_t)m.world);
    int64_t total = 0;
    for (int r = 0; r < m.world; r++) {
        if (all[(size_t)r * 2] != VDL_OK) {
            if (rc != VDL_OK) throw Error(rc, own);
            throw Error(VDL_ERR_UNSUPPORTED, "sharded run: the scan above the position sets failed on rank " + std::to_string(r) + " (its error is reported there)");
        }
        cnt[(size_t)r] = all[(size_t)r * 2 + 1];
        total += cnt[(size_t)r];
    }
    BufP recv = dev_alloc(c, sizeof(int64_t) * (size_t)std::max<int64_t>(total * ncols, 1));
    std::vector<const int64_t *> from;
    std::vector<int64_t *> to;
    for (int k = 0; k < ncols; k++) { from.push_back((const int64_t *)send->p + (int64_t)k * n_send); to.push_back((int64_t *)recv->p + (int64_t)k * total); }
    all_gather_rows(c, from, to, cnt, c->stream);
    p->shard_keep = recv;
    if (vdl_exchange_finish(c, p, recv->p, total) != VDL_OK) throw Error(VDL_ERR_DEVICE, c->err);
}

static bool fold_route(vdl_ctx *c, vdl_plan *p) {
    int64_t nw = 0;
    const int32_t *ops = nullptr;
    const std::string keep = c->err;
    const bool ok = vdl_plan_partial_spec(p, &nw, &ops) == VDL_OK;
    if (!ok) c->err = keep;
    return ok;
}

}  // namespace eng
}  // namespace vdl

extern "C" {

int vdl_comm_unique_id(void *id_out) {
    if (!id_out) return VDL_ERR_ARG;
    try {
        ncclUniqueId id;
        if (rccl().GetUniqueId(&id) != kNcclSuccess) return VDL_ERR_DEVICE;
        std::memcpy(id_out, &id, sizeof id);
        return VDL_OK;
    } catch (const Error &e) {
        std::fprintf(stderr, "vdl_comm_unique_id: %s\n", e.what());
        return e.code;
    }
}

int vdl_comm_init(vdl_ctx *c, int rank, int world, const void *id) {
    if (!c || !id || world < 1 || rank < 0 || rank >= world || world > kMaxExWorld) return VDL_ERR_ARG;
    return guard(c, [&] {
        need_device(c);
        auto m = std::make_shared<CommState>();
        m->kind = CommState::RCCL; m->rank = rank; m->world = world;
        ncclUniqueId uid;
        std::memcpy(&uid, id, sizeof uid);
        RCCL_CHECK(rccl().CommInitRank(&m->comm, world, uid, rank));
        int count = 0;
        RCCL_CHECK(rccl().CommCount(m->comm, &count));
        if (count != world) throw Error(VDL_ERR_DEVICE, "the RCCL communicator has " + std::to_string(count) + " rank(s), " + std::to_string(world) + " expected");
        HIP_CHECK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
        c->comm = m;
    });
}

int vdl_comm_init_host(vdl_ctx *c, int rank, int world, const vdl_comm_host *transport) {
    if (!c || !transport || !transport->all_gather || !transport->all_to_all || world < 1 || rank < 0 || rank >= world || world > kMaxExWorld) return VDL_ERR_ARG;
    return guard(c, [&] {
        auto m = std::make_shared<CommState>();
        m->kind = CommState::HOST; m->rank = rank; m->world = world; m->host = *transport;
        c->comm = m;
    });
}

int vdl_comm_info(const vdl_ctx *c, int *rank, int *world, const char **transport) {
    if (!c || !c->comm) return VDL_ERR_ARG;
    if (rank) *rank = c->comm->rank;
    if (world) *world = c->comm->world;
    if (transport) *transport = c->comm->kind == CommState::RCCL ? "rccl" : "host";
    return VDL_OK;
}

void vdl_comm_free(vdl_ctx *c) {
    if (!c || !c->comm) return;
    if (c->device >= 0) { (void)hipSetDevice(c->device); (void)hipDeviceSynchronize(); }
    c->comm.reset();
}

/* which route vdl_run_sharded takes for this plan and placement: "fold" (partial words merged: every rank ends with the whole
 * answer), "set" (a semi-join set merged, scans over replicated tables: every rank ends with the whole answer), "exchange" (rows
 * travel by key range: the ranks' outputs concatenate in rank order), "front" (the fused front's survivors are all-gathered and
 * the statements above it run on every rank: every rank ends with the whole answer), "replicate" (no other route: the sharded
 * table's columns the plan loads are all-gathered once and the whole query runs on every rank); VDL_ERR_UNSUPPORTED with the reason
 * when there is none (VDL_NO_REPLICATE_ROUTE) */
int vdl_plan_sharded_route(vdl_ctx *c, vdl_plan *p, const char **route, int *replicated) {
    if (!c || !p) return VDL_ERR_ARG;
    const char *name = nullptr;
    bool semi = false;
    if (p->use_fusion && p->fused.ok) for (const PreludeItem &it : p->fused.prelude) semi |= it.kind == PreludeItem::SEMI_BITMAP;
    if (semi) {
        const std::string why = semi_route_refusal(p);
        if (!why.empty()) { c->err = "no sharded route: " + why; return VDL_ERR_UNSUPPORTED; }
        name = "set";
    } else if (fold_route(c, p)) {
        name = "fold";
    } else {
        int ncols = 0;
        if (p->sharded_table.empty()) { c->err = "name the row-sharded table first (vdl_plan_set_sharded_table)"; return VDL_ERR_ARG; }
        p->ex_allow_folds = true;                              // (vdl_run_sharded merges global folds beside the Partition)
        const int rc = vdl_exchange_spec(p, p->sharded_table.c_str(), &ncols);
        p->ex_allow_folds = false;
        if (rc == VDL_OK) name = "exchange";
        else {
            const std::string why_not_exchange = c->err;
            const std::string why = front_route_refusal(p);
            std::string why_not_chain;
            if (why.empty()) name = "front";
            else if (chain_route(p, &why_not_chain)) { c->err.clear(); name = "chain"; }
            else if (getenv("VDL_NO_REPLICATE_ROUTE")) { c->err = why_not_exchange + "; no front route either: " + why + "; no chain route: " + why_not_chain; return rc; }
            else {
                // the last resort: the table's columns gathered once, the whole query on every rank -- the reasons stay readable
                c->err.clear();
                name = "replicate";
            }
        }
    }
    if (route) *route = name;
    if (replicated) *replicated = std::strcmp(name, "exchange") != 0;
    return VDL_OK;
}

int vdl_run_sharded_begin(vdl_ctx *c, vdl_plan *p, int slot) {
    if (!c || !p) return VDL_ERR_ARG;
    return guard(c, [&] {
        if (!fold_route(c, p)) throw Error(VDL_ERR_UNSUPPORTED, "vdl_run_sharded_begin serves plans whose outputs are folds (partial words); plans with a "
                                                               "Partition exchange rows and run through vdl_run_sharded");
        sharded_begin(c, p, slot);
    });
}

int vdl_run_sharded_end(vdl_ctx *c, vdl_plan *p, int slot) {
    if (!c || !p) return VDL_ERR_ARG;
    const int rc = vdl_finalize_end(c, p, slot);
    const int rs = guard(c, [&] { sharded_check_status(p, slot); });      // (a failed peer outranks whatever the merged zeros gave)
    return rs != VDL_OK ? rs : rc;
}

int vdl_run_sharded(vdl_ctx *c, vdl_plan *p) {
    if (!c || !p) return VDL_ERR_ARG;
    if (p->use_fusion && p->fused.ok) {                       // a fused plan with a semi-join set: the sets are merged across the ranks
        bool semi = false;
        for (const PreludeItem &it : p->fused.prelude) semi |= it.kind == PreludeItem::SEMI_BITMAP;
        if (semi) {
            if (!c->comm || c->comm->world == 1) return vdl_run(c, p);
            const std::string why = semi_route_refusal(p);
            if (!why.empty()) { c->err = "this fused plan builds a semi-join set and has no sharded route here: " + why; return VDL_ERR_UNSUPPORTED; }
            return guard(c, [&] { sharded_semi(c, p); });
        }
    }
    int rc = guard(c, [&] {
        if (fold_route(c, p)) { sharded_begin(c, p, 0); return; }
        if (c->comm && !p->sharded_table.empty()) {
            // rows travel by key range when the plan allows it; otherwise, if its work on the sharded table is a fused front, the
            // survivors are gathered and the rest runs on every rank (the "front" route)
            int ncols = 0;
            const std::string keep = c->err;
            p->ex_allow_folds = true;
            const bool exchange_ok = vdl_exchange_spec(p, p->sharded_table.c_str(), &ncols) == VDL_OK;
            p->ex_allow_folds = false;
            if (!exchange_ok) {
                const bool front = front_route_refusal(p).empty();
                const bool chain = !front && chain_route(p);
                if (front || chain || !getenv("VDL_NO_REPLICATE_ROUTE")) {
                    c->err = keep;
                    if (c->comm->world > 1 || getenv("VDL_FRONT_ROUTE_ALWAYS")) { if (front) sharded_front(c, p); else if (chain) sharded_chain(c, p); else sharded_replicate(c, p); }      // (the switch: tests send a one-rank communicator through the collectives)
                    else if (vdl_run(c, p) != VDL_OK) throw Error(VDL_ERR_DEVICE, c->err);      // one rank holds the whole table
                    return;
                }
            }
            c->err = keep;
        }
        sharded_exchange(c, p);
    });
    if (rc != VDL_OK) return rc;
    return fold_route(c, p) ? vdl_run_sharded_end(c, p, 0) : VDL_OK;
}

/* The merge of the gathered partial words on the HOST: the same per-word rule the device kernel applies (merge_word,
 * vdl_kernels.h), exported so that hosts and CPU tests can check a transport without a GPU. */
int vdl_comm_merge_host(int world, int64_t n_words, const int32_t *ops, const int64_t *gathered, int64_t stride_words, int64_t *out, int64_t *status_out) {
    if (stride_words == 0) stride_words = 2 * n_words;
    if (world < 1 || n_words < 0 || stride_words < 2 * n_words || (n_words && (!ops || !gathered || !out))) return VDL_ERR_ARG;
    if (status_out && (stride_words < 2 * n_words + 1 || !gathered)) return VDL_ERR_ARG;
    for (int64_t i = 0; i < n_words; i++) out[i] = merge_word(gathered, world, n_words, ops[i], i, stride_words);
    if (status_out) {                                           // as k_merge_words: the first rank whose local phase failed
        status_out[0] = 0; status_out[1] = -1;
        for (int r = world - 1; r >= 0; r--) {
            const int64_t x = gathered[(int64_t)r * stride_words + 2 * n_words];
            if (x != 0) { status_out[0] = x; status_out[1] = r; }
        }
    }
    return VDL_OK;
}

}  // extern "C"

/*
 * vdl.h -- C ABI of libvdl, the MI355X-native execution engine for the textual VDL
 * (Voodoo vector-operator dataflow) printed by orm011/mplan2vdl.
 *
 * What this boundary replaces.  The reference has no in-process executor: its VDL text
 * leaves the process on stdout (/root/reference/src/MainFuns.hs:157) and is POSTed to
 * an external Voodoo server, whose JSON reply is decoded by resolve.py
 * (/root/reference/eval_query.sh:18-26, /root/reference/resolve.py:8-32).  libvdl is
 * that server's `/voodoo/.../run` hop as a library:
 *
 *     request  body  = VDL text (grammar: /root/reference/src/Vdl.hs:410-477;
 *                      " ;; metadata" suffixes as stripped by eval_query.sh:20 are ignored)
 *     response body  = {"results": {"tmpN": {".<name>": [ints]}}, "timings": {label: usec}}
 *                      -> vdl_output() / vdl_timing() below, one entry per MaterializeCompact.
 *
 * A Haskell host binds these with `foreign import ccall` (stub in INTEGRATION.md); the
 * CLI `vdlrun` and the Python package `mplan2vdl_amd` are the callers exercised here.
 *
 * Conventions: plain C types only, no callbacks (except the optional host transport of vdl_comm_init_host); every call returns VDL_OK (0) or a
 * VDL_ERR_* code and leaves a message for vdl_last_error(); pointers returned by
 * vdl_output()/vdl_timing()/vdl_plan_describe() are borrowed and stay valid until the
 * plan is run again or freed.  One context per process and GPU; calls on one context
 * are not re-entrant (the reference's model is one query per request, no shared state).
 * There is NO CPU fallback: every run call needs a HIP device and fails with
 * VDL_ERR_DEVICE without one.
 */
#ifndef VDL_H
#define VDL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vdl_ctx  vdl_ctx;   /* device, stream, column catalog, memory pool   */
typedef struct vdl_plan vdl_plan;  /* parsed program + fused execution plan + results */

enum {
    VDL_OK              = 0,
    VDL_ERR_PARSE       = 1,  /* malformed VDL text (line number in vdl_last_error)          */
    VDL_ERR_COLUMN      = 2,  /* Load of a column that is not in the catalog                  */
    VDL_ERR_UNSUPPORTED = 3,  /* operator / pattern outside the implemented set               */
    VDL_ERR_DEVICE      = 4,  /* no HIP device, or a HIP call failed                          */
    VDL_ERR_ARG         = 5,  /* bad argument                                                 */
    VDL_ERR_SHAPE       = 6,  /* operand lengths / field names do not fit                     */
    VDL_ERR_NOMEM       = 7
};

/* partial-state merge operators for sharded execution */
enum {
    VDL_REDUCE_NONE = 0, VDL_REDUCE_SUM = 1, VDL_REDUCE_MIN = 2, VDL_REDUCE_MAX = 3,
    /* FoldChoose of a grouped plan: the word holds the group's smallest GLOBAL row id after the local
     * phase.  Merge = all-reduce MIN, then vdl_resolve_first() (the rank owning that row substitutes the
     * column value, every other rank 0), then all-reduce SUM of the same words. */
    VDL_REDUCE_FIRST = 4
};

/* ---- context ------------------------------------------------------------------ */

/* device >= 0 binds that HIP device and creates the engine stream; device < 0 makes a
 * host-only context that can parse, plan and describe but not run (used by CPU tests). */
int  vdl_open(vdl_ctx **out, int device);
void vdl_close(vdl_ctx *ctx);
const char *vdl_last_error(const vdl_ctx *ctx);
const char *vdl_version(void);

/* Run on the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream).  The handle is
 * used as given: 0 / NULL means the legacy default stream (what torch uses unless told otherwise),
 * so that collectives and tensor ops issued by the caller order against the engine's kernels.
 * vdl_use_own_stream() goes back to the engine's private non-blocking stream.  Both drain the stream being left
 * first (the engine's buffers are protected by stream order), so switch streams between queries, not inside one. */
int  vdl_set_stream(vdl_ctx *ctx, void *hip_stream);
int  vdl_use_own_stream(vdl_ctx *ctx);

/* ---- column catalog ("Load table.col", /root/reference/src/Vdl.hs:161-168,419-420) --
 * Columns are contiguous little-endian signed integers of elem_bytes in {1,2,4,8}
 * (storage widths /root/reference/tests/tpch10noorder/storage.csv:188-208), resident in
 * HBM.  `name` is the key path as printed after Load, e.g. "lineitem.l_shipdate". */
int  vdl_register_column(vdl_ctx *ctx, const char *name, const void *dev_ptr,
                         int elem_bytes, int64_t nrows);               /* borrowed device memory */
int  vdl_upload_column(vdl_ctx *ctx, const char *name, const void *host_ptr,
                       int elem_bytes, int64_t nrows);                 /* engine-owned copy, H2D   */
/* Synthetic column generated in place in HBM by the counter-based generator of
 * v(row) = add + mul*(lo + splitmix64(seed ^ fnv1a(name)*PHI ^ row) % (hi-lo+1)),
 * rows [row0, row0+nrows). */
int  vdl_generate_column(vdl_ctx *ctx, const char *name, int elem_bytes, int64_t row0,
                         int64_t nrows, uint64_t seed, int64_t lo, int64_t hi,
                         int64_t mul, int64_t add);
int  vdl_drop_column(vdl_ctx *ctx, const char *name);
int  vdl_column_info(const vdl_ctx *ctx, const char *name, int *elem_bytes, int64_t *nrows,
                     const void **dev_ptr);
/* Copy a catalog column back to the host (tests, debugging). */
int  vdl_download_column(vdl_ctx *ctx, const char *name, void *host_ptr, size_t bytes);

/* ---- plans --------------------------------------------------------------------- */

/* Parse the VDL text, check it, and build the execution plan (operator fusion included).
 * Needs no device.  Both output formats of the compiler are accepted: the default VDL lines
 * (/root/reference/src/Vdl.hs:410-453) and the --vliteformat lines (Vdl.hs:370-408,455-475; recognised by
 * their Output statements).
 * One point where a program's meaning is narrower than its text: Scatter(src, fold, pos) with the SAME position twice
 * leaves one of the two values there, which one is unspecified (the oracle keeps the later slot's, the kernels whichever
 * store lands last).  Every Scatter the compiler emits has unique positions -- Partition ranks, filtered row ids
 * (/root/reference/src/Vlite.hs:508,1058-1059,1267-1275) -- and the parity tests only build such programs. */
int  vdl_parse(vdl_ctx *ctx, const char *vdl_text, size_t len, vdl_plan **out);
void vdl_plan_free(vdl_plan *plan);

/* Human-readable plan: one line per execution step ("scan ...", "op 17 Greater ..."). */
const char *vdl_plan_describe(const vdl_plan *plan);
/* 1 if every output of the plan comes from fused scan kernels (no per-operator steps). */
int  vdl_plan_is_fused(const vdl_plan *plan);
/* Force per-operator execution (no fusion) for this plan: used by parity tests to check
 * every operator kernel against the oracle on programs that would otherwise fuse. */
int  vdl_plan_set_fusion(vdl_plan *plan, int enabled);

/* Run-time specialisation of a fused plan's multi-aggregate scans: the scan kernels' own device code compiled by hiprtc with
 * this plan's descriptor (column kinds, filters, group key, aggregate terms, conditions) as compile-time constants, at the
 * next run after it is switched on -- seconds, once per distinct scan per process (kept under $VDL_JIT_CACHE when that
 * names a directory).  Off by default (VDL_JIT=1 in the environment switches it on for every plan parsed afterwards).
 * enabled = 2 (VDL_JIT=2) also tunes: at the first run each specialised scan is built with 2, 3, 4 and 6 row pairs per
 * lane and the quickest of a few timed launches over the real columns stays (a few seconds more, once).  A scan
 * whose specialisation does not build runs on the precompiled kernels; vdl_plan_jit_note says which did and which did not.
 * vdl_plan_jit_check builds the specialised kernels against the columns registered now without loading or running them
 * (no GPU needed): VDL_OK, or VDL_ERR_UNSUPPORTED with the compiler's message in vdl_last_error.
 * (The reference hands its text to an engine that generates code per program: /root/reference/README.md:57.) */
int  vdl_plan_set_jit(vdl_plan *plan, int enabled);
const char *vdl_plan_jit_note(const vdl_plan *plan);
int  vdl_plan_jit_check(vdl_ctx *ctx, vdl_plan *plan);

/* Execute: binds Loads to the catalog, runs all kernels, copies the MaterializeCompact
 * outputs to the host and synchronises. */
int  vdl_run(vdl_ctx *ctx, vdl_plan *plan);

int  vdl_n_outputs(const vdl_plan *plan);
/* k-th output in program order: `name` is the output field (resolve.py:64-78 splits it on
 * "__"), `tmp` the "tmpN" result key (N = id of the MaterializeCompact line). */
int  vdl_output(const vdl_plan *plan, int k, const char **name, const char **tmp,
                const int64_t **vals, size_t *n);
/* Results that stay in HBM.  With device outputs enabled, an output of 65536 values or more is not copied to the
 * host: vdl_output reports its length with *vals == NULL and vdl_output_device hands out the device pointer (int64
 * values, owned by the plan until it is run again or freed; the run has completed on the context's stream when
 * vdl_run returns).  Smaller outputs stay host-side (*dev_vals == NULL).  Q3 at SF100 returns 4 x 13.9M values:
 * 445 MB over PCIe is a quarter of the run. */
int  vdl_plan_set_device_outputs(vdl_plan *plan, int enabled);
int  vdl_output_device(const vdl_plan *plan, int k, const int64_t **dev_vals, size_t *n);
int  vdl_n_timings(const vdl_plan *plan);
int  vdl_timing(const vdl_plan *plan, int k, const char **label, double *usec);

/* Debugging aid for parity work.  With tracing on, a vdl_run that goes statement by statement (plan not fused, or
 * vdl_plan_set_fusion(plan, 0)) keeps a host copy of every statement's vector as it stood right after the statement:
 * n slots, vals[i] and ok[i] (1 = the slot holds a value, 0 = EPS; vals is 0 there).  `form` names the engine's
 * internal representation ("dense", "sparse", "range", "onehot", ...).  vals / ok are NULL for statements whose
 * evaluation was deferred at that point (fused expression trees, lazy gathers) and for vectors of more than 2^22
 * slots.  Entries are in execution order and stay valid until the plan runs again or is freed. */
int  vdl_plan_set_trace(vdl_plan *plan, int enabled);
int  vdl_n_traced(const vdl_plan *plan);
int  vdl_traced(const vdl_plan *plan, int k, int *node_id, const char **form, int64_t *n,
                const int64_t **vals, const uint8_t **ok);

/* Per-kernel device time of the last vdl_run()/vdl_run_local(), measured with HIP events
 * on the stream the kernels were launched on; enabled with vdl_plan_set_profiling(). */
int  vdl_plan_set_profiling(vdl_plan *plan, int enabled);
/* Rows scanned and algorithmic bytes read by the dominant (fused scan) kernel of the last
 * run, and its device time in microseconds (0 if profiling was off). */
int  vdl_plan_scan_stats(const vdl_plan *plan, int64_t *rows, int64_t *algo_bytes, double *usec);
/* HBM bytes ONE launch of that kernel moves over the columns registered now (measurement; call after a run, outside any
 * timed region).  A scan that reads every column with the tile moves its algorithmic bytes.  A scan specialised to read
 * late (vdl_plan_set_jit: staged reads) moves the eager columns in full plus 128 bytes for every cache line of a late
 * column in which some row was still in when the column was read -- the memory side fetches whole 128-byte lines
 * (tools/ubench/fetch_calib.hip) -- and that number is COUNTED: a census build of the same kernel form runs once.
 * `detail` (may be null): "column=bytes ..." text, valid until the next call. */
int  vdl_plan_scan_traffic(vdl_ctx *ctx, vdl_plan *plan, int64_t *bytes_moved, const char **detail);

/* ---- sharded execution: one process per GPU, columns sharded by row range --------
 * A plan whose outputs are global folds keeps its mergeable state in `n_words` int64
 * words, each tagged with a VDL_REDUCE_* operator.  Each rank runs the local phase over
 * its row range, the caller merges the words across ranks (RCCL all-reduce through
 * torch.distributed, one call per operator class), and every rank finalises. */
int  vdl_plan_partial_spec(const vdl_plan *plan, int64_t *n_words, const int32_t **reduce_ops);
/* Local phase, asynchronous on the context stream; dev_partials = caller-owned device
 * buffer of n_words int64 (fully overwritten). */
int  vdl_run_local(vdl_ctx *ctx, vdl_plan *plan, void *dev_partials);
/* After the merge: produce the outputs from the (merged) words; synchronises. */
int  vdl_finalize(vdl_ctx *ctx, vdl_plan *plan, const void *dev_partials);
/* Placement for plans that do not fuse: `table` is split by rows over the ranks, every other table is replicated.
 * With it, vdl_plan_partial_spec / vdl_run_local / vdl_finalize also serve general plans whose outputs hang off
 * GLOBAL folds over that table (a join followed by an ungrouped aggregate: TPC-H Q14, Q19): three words per fold
 * {value, first slot, count}, everything below the folds checked to be row-local, the statements above them run on
 * the merged scalars.  VDL_ERR_UNSUPPORTED with the reason otherwise.  Also what vdl_exchange_spec records. */
int  vdl_plan_set_sharded_table(vdl_plan *plan, const char *table);
/* Global index of this rank's first row (default 0); row ids in VDL_REDUCE_FIRST words are global. */
int  vdl_plan_set_row_offset(vdl_plan *plan, int64_t row0);
/* Second merge phase of VDL_REDUCE_FIRST words (see the enum); asynchronous on the context stream. */
int  vdl_resolve_first(vdl_ctx *ctx, vdl_plan *plan, void *dev_partials);
/* Pipelined form (two slots, 0 and 1): `begin` enqueues the copy of the merged words to a pinned host
 * slot and returns at once; `end` waits for that copy only -- younger launches on the stream keep
 * running -- and produces the outputs.  Lets a driver overlap the host side of query k with the
 * kernels of query k+1. */
int  vdl_finalize_begin(vdl_ctx *ctx, vdl_plan *plan, const void *dev_partials, int slot);
int  vdl_finalize_end(vdl_ctx *ctx, vdl_plan *plan, int slot);

/* ---- sharded Partition (joins / sparse GROUP BY, e.g. TPC-H Q3): row exchange -------------------
 * For plans that are not fused but whose outputs depend on the sharded table only through
 * Scatter(x, _, Partition(key, RangeC min cnt 1)) -- the group-by lowering of
 * /root/reference/src/Vlite.hs:1056-1060,1082-1098 -- every rank
 *   1. vdl_exchange_begin : runs the statements up to the partition key and the scattered vectors on
 *      its own rows and reports how many rows go to each rank (rank r owns the keys of
 *      [min + r*cnt/world, min + (r+1)*cnt/world)); rows keep their order inside each destination;
 *   2. vdl_exchange_pack  : writes the n_send = sum(counts) rows, grouped by destination, into a
 *      caller-owned device buffer of n_columns x n_send int64 (column c starts at c * n_send):
 *      key, scattered vectors, one validity word;
 *   3. (caller) all-to-all of every column over RCCL (torch.distributed.all_to_all_single with the
 *      counts as split sizes);
 *   4. vdl_exchange_finish: runs Partition / Scatter / Fold / MaterializeCompact on the received
 *      rows (n_columns x n_recv int64, same layout).  The outputs of rank 0, 1, ... concatenate to the
 *      unsharded result (keys ascend across ranks; stability makes FoldChoose pick the same row).
 * Dimension tables must be replicated on every rank; only the partitioned table is sharded.
 * vdl_exchange_pack returns when the send buffer is complete; the received rows must be complete
 * when vdl_exchange_finish is called (synchronise the collective's stream first unless the engine
 * runs on that stream, vdl_set_stream).
 * vdl_exchange_spec checks the structure; with sharded_table != NULL ("lineitem") it also verifies
 * that everything below the Partition is row-local over that table (element-wise operators,
 * constants, Gathers out of replicated vectors), lets the statements above the Partition read the
 * replicated tables' columns (the plan remembers the placement for vdl_exchange_begin), and returns
 * VDL_ERR_UNSUPPORTED with the reason
 * otherwise.  world <= 128. */
int  vdl_exchange_spec(const vdl_plan *plan, const char *sharded_table, int *n_columns);
int  vdl_exchange_begin(vdl_ctx *ctx, vdl_plan *plan, int world, int64_t *counts_host /* world */);
int  vdl_exchange_pack(vdl_ctx *ctx, vdl_plan *plan, void *dev_send);
int  vdl_exchange_finish(vdl_ctx *ctx, vdl_plan *plan, const void *dev_recv, int64_t n_recv);

/* ---- multi-GPU behind this boundary: one process per GPU, the context owns the communicator --------------
 * (SURVEY.md section 8(b),(e).)  A Haskell host -- or vdlrun --gpus N, or bench.py -- starts one process per GPU; each opens
 * its context on its device, holds its row range of the sharded table (and the replicated tables in full), joins the
 * communicator and calls vdl_run_sharded(); the collectives happen inside.
 *
 *   vdl_comm_unique_id : any one rank makes the 128-byte id (ncclGetUniqueId) and the host hands the bytes to the others
 *                        (a file, a pipe, an environment variable, MPI, a torch store: any channel);
 *   vdl_comm_init      : RCCL communicator over xGMI on the context's device (librccl is opened here, not at load time);
 *   vdl_comm_init_host : instead of RCCL, collectives supplied by the caller over HOST memory (hosts that already have
 *                        MPI / gloo; the tests' in-process stand-in).  The engine stages through pinned buffers.  These two
 *                        callbacks are the only ones in this interface; both are called by every rank, in the same order.
 *
 * vdl_run_sharded picks the route from the plan:
 *   outputs = global / dense-domain grouped folds (vdl_plan_partial_spec succeeds: fused plans, or general plans after
 *   vdl_plan_set_sharded_table): local phase -> ONE all-gather of the partial words -> merge kernel -> finalise; every rank
 *   ends with the full result.  FoldChoose words travel as (global row id, value) pairs: no second round.
 *   plans with a Partition (vdl_plan_set_sharded_table names the row-sharded table): local phase -> ONE all-gather of
 *   {status, rows per destination} -> ONE grouped send / receive of all columns -> local tail; rank r ends with the
 *   groups of key range r, and the ranks' outputs concatenate in rank order to the unsharded result.
 * (more routes: below, at vdl_plan_sharded_route)
 * vdl_run_sharded_begin / _end split the fold route for pipelined callers (slots 0 / 1): `begin` queues the scan on the
 * engine stream and merge + copy-out behind it on the communication stream and returns; `end` waits for that slot's copy
 * only, so the collective of query k hides behind the scan of query k+1. */
#define VDL_COMM_ID_BYTES 128
typedef struct vdl_comm_host {
    void *user;
    /* every rank contributes `bytes` from `send`; `recv` (world * bytes) gets rank r's block at r * bytes.  0 = ok. */
    int (*all_gather)(void *user, const void *send, void *recv, size_t bytes);
    /* rank-major blocks: send_bytes[r] bytes go to rank r, recv_bytes[r] bytes arrive from rank r.  0 = ok. */
    int (*all_to_all)(void *user, const void *send, const size_t *send_bytes, void *recv, const size_t *recv_bytes);
} vdl_comm_host;
int  vdl_comm_unique_id(void *id_out /* VDL_COMM_ID_BYTES */);
int  vdl_comm_init(vdl_ctx *ctx, int rank, int world, const void *id /* VDL_COMM_ID_BYTES */);
int  vdl_comm_init_host(vdl_ctx *ctx, int rank, int world, const vdl_comm_host *transport);
int  vdl_comm_info(const vdl_ctx *ctx, int *rank, int *world, const char **transport /* "rccl" | "host" */);
void vdl_comm_free(vdl_ctx *ctx);                       /* also done by vdl_close */
/* A third route:
 *   fused plans with a semi-join set (EXISTS / IN, TPC-H Q4) whose source table is the sharded one and whose scans read replicated
 *   tables: every rank builds the set from its rows -> ONE all-gather of the sets -> OR kernel -> the scans run everywhere
 *   against the complete set; every rank ends with the full result.
 * A fourth, for plans the exchange cannot serve (two Partitions, folds over grouped results: TPC-H Q16) whose work on the sharded
 *   table is a fused front (filter + projection in one scan): every rank runs the front over its rows -> all-gather of the row
 *   counts -> all-gather of {status, survivors} -> ONE grouped send / receive of the survivors' vectors to every rank (rank after
 *   rank = row order) -> the statements above the front run everywhere on the complete vectors; every rank ends with the full
 *   result.  Tried after the exchange; the row-id conditions of the front count from the table's first row.
 * A fifth, the "chain" (TPC-H Q18: a GROUP BY over ALL rows of the sharded table whose groups only feed position sets --
 *   Scatter(constant, size, a value of the group): the semi-join set of `in (select .. group by .. having ..)` -- and a rest that reads the
 *   sets and the table a second time): the rows travel to the owners of their key range as on the exchange route and every owner runs the
 *   GROUP BY on complete groups -> all-gather of {status, positions found, length} + ONE grouped send / receive per set: every rank
 *   builds the same sets -> the rest runs with the sets given: per-row work on each rank's OWN rows, all-gather of {status, rows} and
 *   ONE grouped send / receive of the rows that reach the next Partition (rank after rank = row order), the tail on every rank; every
 *   rank ends with the full result (VDL_NO_CHAIN_ROUTE=1 switches it off).
 * The last resort, for a plan none of the five serves: the sharded table's columns the plan loads are all-gathered ONCE per
 *   catalog state into plan-owned buffers, and every rank runs the whole query over them -- the query itself does not scale, later
 *   runs move nothing, every rank ends with the full result ("replicate"; VDL_NO_REPLICATE_ROUTE=1 turns it into the refusal with the
 *   reasons).
 * vdl_plan_sharded_route tells which route a plan takes ("fold" | "set" | "exchange" | "front" | "chain" | "replicate") and whether every rank
 * ends with the whole answer (replicated = 1) or with its slice (0: concatenate the ranks' outputs in rank order). */
int  vdl_plan_sharded_route(vdl_ctx *ctx, vdl_plan *plan, const char **route, int *replicated);
int  vdl_run_sharded(vdl_ctx *ctx, vdl_plan *plan);     /* results through vdl_output as after vdl_run */
int  vdl_run_sharded_begin(vdl_ctx *ctx, vdl_plan *plan, int slot);
int  vdl_run_sharded_end(vdl_ctx *ctx, vdl_plan *plan, int slot);
/* The merge rule of the gathered partial words on the host (what the device kernel computes): `gathered` holds, per rank, a block
 * of `stride_words` int64 -- n_words words, the same words with VDL_REDUCE_FIRST entries resolved to values and, in the blocks
 * vdl_run_sharded itself gathers, ONE status word of the rank's local phase (stride_words = 2 * n_words + 1; pass 0 for bare blocks
 * of 2 * n_words).  status_out (may be NULL; needs the status word): [0] = the first non-zero status over the ranks, [1] = that
 * rank (-1: every local phase succeeded).  For hosts / tests that want to check a transport without a GPU. */
int  vdl_comm_merge_host(int world, int64_t n_words, const int32_t *ops, const int64_t *gathered, int64_t stride_words, int64_t *out,
                         int64_t *status_out);

#ifdef __cplusplus
}
#endif
#endif /* VDL_H */

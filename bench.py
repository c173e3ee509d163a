#!/usr/bin/env python3
"""bench.py -- TPC-H Q6 at SF100 through the MI355X VDL engine (BASELINE.json's metric).

One "step" = one execution of the Q6 VDL program (tests/golden/q6.vdl, the text
`./tpchrun tests/tpch10noorder tests/tpch10noorder/06.sql.mplan` prints) over the synthetic
lineitem columns already resident in HBM: fused scan kernel, partial fold, (N > 1: RCCL
all-reduce of the partial words), finalisation and the 8-byte answer copied to the host.
N > 1 shards the SF100 rows by contiguous row range, one process per GPU (strong scaling,
as BASELINE.json config "Q6 SF100 row-range sharded across 8xMI355X").

Prints ONE JSON line (rank 0).  `value` = rows/s over all GPUs; kernel time = mean fused-scan
time from HIP events on the launch stream, recorded by libvdl around the kernel inside the
timed region.  `roofline.achieved` / `frac` are on the bytes the kernel MOVED
(vdl_plan_scan_traffic: counted, never above the peak); the 28 B/row figure of SURVEY.md 8(d) is `algorithmic_equivalent_GBps`,
and `read_everything_kernel` is the kernel that moves exactly those bytes, timed in the same run.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_ACHIEVABLE_GBPS = 6300.0   # same guide, "HBM [CDNA4]": 8 TB/s peak (spec); about 6.3 TB/s achievable


Q1_SQL_ORDER = ["lineitem.l_shipdate", "lineitem.l_returnflag", "lineitem.l_linestatus", "lineitem.l_quantity",
                "lineitem.l_extendedprice", "lineitem.l_discount", "lineitem.l_tax"]
Q1_OUTPUTS = ["l_returnflag__lineitem__l_returnflag", "l_linestatus__lineitem__l_linestatus", "sum_qty", "sum_base_price",
              "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc", "count_order"]


def q1_matches_sql(results, total_rows, threads, report=None):
    """Bit-exact check of a full-size Q1 answer: the SQL-semantics loop (oracle/vdl_oracle.c:orc_sql_q1_generated) over
    the regenerated rows on the host cores.  Test infrastructure, always outside any timed region."""
    import oracle
    from mplan2vdl_amd import datagen

    specs = [(datagen.SEED, datagen.col_id(c), datagen.LINEITEM[c].lo, datagen.LINEITEM[c].hi,
              datagen.LINEITEM[c].mul, datagen.LINEITEM[c].add) for c in Q1_SQL_ORDER]
    tab = oracle.sql_q1_generated(specs, 0, total_rows, threads=threads)
    flat = {list(v.keys())[0][1:]: list(v.values())[0] for v in results.values()}
    ok = True
    for j, nm in enumerate(Q1_OUTPUTS):
        if flat.get(nm) != [int(x) for x in tab[:, j]]:
            ok = False
            if report:
                report("  %s: gpu %r\n  %s: cpu %r" % (nm, flat.get(nm), nm, [int(x) for x in tab[:, j]]))
    return ok


def q3_matches_sql(eng, res):
    """Q3's SQL (tests/golden/tpch10noorder/03.sql.mplan:1-19) evaluated with numpy over the columns in HBM (downloaded):
    a checker, outside every timed region.  Result rows are compared as a sorted list (the query carries no ORDER BY)."""
    import numpy as np

    c = {n: eng.download(n) for n in ("customer.c_mktsegment", "orders.o_orderdate", "orders.o_shippriority", "orders.orders_customer",
                                      "lineitem.lineitem_orders", "lineitem.l_orderkey", "lineitem.l_shipdate", "lineitem.l_extendedprice",
                                      "lineitem.l_discount")}
    order_ok = (c["orders.o_orderdate"] < 728732) & (c["customer.c_mktsegment"][c["orders.orders_customer"]] == 16)     # date '1995-03-15', 'BUILDING'
    l_ord = c["lineitem.lineitem_orders"]
    rows = np.nonzero((c["lineitem.l_shipdate"] > 728732) & order_ok[l_ord])[0]
    okey = c["lineitem.l_orderkey"][rows].astype(np.int64)
    uniq, first, inv = np.unique(okey, return_index=True, return_inverse=True)      # one order = one (orderkey, date, priority) group
    rev = np.zeros(len(uniq), np.int64)
    np.add.at(rev, inv.reshape(-1), c["lineitem.l_extendedprice"][rows] * (100 - c["lineitem.l_discount"][rows]))
    fr = l_ord[rows[first]]
    want = np.stack([uniq, rev, c["orders.o_orderdate"][fr].astype(np.int64), c["orders.o_shippriority"][fr].astype(np.int64)], axis=1)
    got = np.stack([np.asarray(res["tmp93"][".l_orderkey__lineitem__l_orderkey"]), np.asarray(res["tmp110"][".revenue"]),
                    np.asarray(res["tmp115"][".o_orderdate__orders__o_orderdate"]), np.asarray(res["tmp120"][".o_shippriority__orders__o_shippriority"])], axis=1)
    return bool(np.array_equal(got[np.argsort(got[:, 0])], want[np.argsort(want[:, 0])]))


Q3_ORDERS = {"sf0.01": 15000, "sf1": 1500000, "sf10": 15000000, "sf100": 150000000}      # lineitem = 4 x orders (datagen.register_q3_columns), customer = orders / 10


def q3_program(n_orders):
    """Q3's VDL for a catalog of that size: the committed fixture up to SF10 (compiled from 03.sql.mplan against the reference's SF10
    metadata: 2^38 group-key domain); beyond it the same plan compiled by the front-end restatement against those bounds scaled
    like TPC-H scales (2^42 at SF100), as `./tpchrun` over an SF100 export would print it."""
    if n_orders <= 15000000:
        return open(os.path.join(ROOT, "tests", "golden", "q3.vdl")).read()
    from mplan2vdl_amd import catalog, frontend
    meta = os.path.join(ROOT, "tests", "golden", "tpch10noorder")
    factor = -(-n_orders // 15000000)
    return frontend.compile_plan(open(os.path.join(meta, "03.sql.mplan")).read(), catalog.tpch_scaled_config(frontend.load_metadata(meta), factor))


def q3_outputs(res):
    """the four result columns of a Q3 reply by field name (statement numbers differ between compilations)"""
    flat = {list(v.keys())[0][1:]: list(v.values())[0] for v in res.values()}
    return (flat["l_orderkey__lineitem__l_orderkey"], flat["revenue"], flat["o_orderdate__orders__o_orderdate"], flat["o_shippriority__orders__o_shippriority"])


def q3_checksums_torch(eng, dev):
    """Q3's SQL (tests/golden/tpch10noorder/03.sql.mplan:1-19) over THIS rank's columns as they lie in HBM, evaluated with plain
    torch tensor operations (a checker: nothing of libvdl's kernels; outside every timed region).  Returns int64 checksums of
    the grouped result -- groups, sum of revenue, sum of order keys, sum of order dates, sum of key x revenue (wrapping) -- that
    add up over ranks when no order's lineitems are split between two of them (shard boundaries are multiples of 4 rows)."""
    import torch
    col = lambda name: torch.as_tensor(eng.column_device(name), device=dev)
    order_ok = (col("orders.o_orderdate") < 728732) & (col("customer.c_mktsegment")[col("orders.orders_customer")] == 16)       # date '1995-03-15', 'BUILDING'
    l_ord = col("lineitem.lineitem_orders")
    rows = torch.nonzero((col("lineitem.l_shipdate") > 728732) & order_ok[l_ord]).reshape(-1)
    del order_ok
    okey = col("lineitem.l_orderkey")[rows].to(torch.int64)
    uniq, inv = torch.unique(okey, return_inverse=True)
    rev = torch.zeros(len(uniq), dtype=torch.int64, device=dev)
    rev.index_add_(0, inv, col("lineitem.l_extendedprice")[rows] * (100 - col("lineitem.l_discount")[rows]))
    date = torch.zeros(len(uniq), dtype=torch.int64, device=dev)
    date.scatter_(0, inv, col("orders.o_orderdate")[l_ord[rows]].to(torch.int64))             # (every row of a group carries its order's date)
    return [int(len(uniq)), int(rev.sum()), int(uniq.sum()), int(date.sum()), int((uniq * rev).sum())]


def q3_checksums_of_result(plan_result, dev):
    import torch
    okey, rev, date, prio = (torch.as_tensor(x, device=dev) if not isinstance(x, list) else torch.tensor(x, dtype=torch.int64, device=dev) for x in q3_outputs(plan_result))
    assert int(prio.abs().sum()) == 0 if len(prio) else True
    return [int(len(okey)), int(rev.sum()), int(okey.sum()), int(date.sum()), int((okey * rev).sum())]


def q3_measure(eng, dev, n_orders, li_range, world, steps, warmup, jit, dist=None, sharded=False):
    """Q3 over the columns register_q3_columns builds in place: `steps` executions with the four result columns left in HBM
    (vdl_plan_set_device_outputs), wall time per query; then the same with the results copied to the host (N = 1 only).
    sharded: vdl_run_sharded -- local phase on this rank's lineitem rows (orders co-partitioned), ONE count all-gather, ONE grouped
    send/receive of the surviving rows by key range over RCCL, tail on the received rows."""
    import torch
    from mplan2vdl_amd import datagen

    keep = datagen.register_q3_columns(eng, n_orders, li_range, device=dev, copartition=world > 1)
    plan = eng.parse(q3_program(n_orders))
    if jit != "off":
        plan.set_jit(True)                        # (its projection / dimension scans; nothing to tune: their tile shape is fixed)
    plan.set_device_outputs(True)
    if sharded:
        plan.set_sharded_table("lineitem")
    go = plan.execute_sharded if sharded else plan.execute

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(warmup, 1)):
        go()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        go()
    sync()
    elapsed = time.perf_counter() - t0
    res = plan.collect()["results"]
    got = q3_checksums_of_result(res, dev)
    want = q3_checksums_torch(eng, dev)
    host_ms = None
    if not sharded:
        plan.set_device_outputs(False)
        wall = []
        for _ in range(5):
            torch.cuda.synchronize(); t1 = time.perf_counter(); plan.execute(); torch.cuda.synchronize(); wall.append(time.perf_counter() - t1)
        host_ms = 1e3 * sum(wall[2:]) / len(wall[2:])
    return {"plan": plan, "keep": keep, "elapsed": elapsed, "got": got, "want": want, "host_ms": host_ms, "note": plan.jit_note()}


def secondary_measurements(eng, rows, jit="tune"):
    """Not the headline metric: the other two single-GPU configurations of BASELINE.json on the same box, measured
    after the timed region (Q1 grouped fused scan over the same lineitem rows; Q3 at SF10 through the statement-by-
    statement executor).  Parity for both is the job of tests/; a failure here is reported, not fatal."""
    import torch

    from mplan2vdl_amd import datagen

    also = {}
    try:
        for name in datagen.Q1_COLUMNS:
            if name not in datagen.Q6_COLUMNS:
                eng.generate(datagen.LINEITEM[name], 0, rows)
        q1 = eng.parse(open(os.path.join(ROOT, "tests", "golden", "q1.vdl")).read())
        q1.set_profiling(True)
        if jit != "off":
            q1.set_jit(True, tune=jit == "tune")
        us, wall = [], []
        for k in range(12):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = q1.run(); wall.append(time.perf_counter() - t0)
            us.append(next(v for lbl, v in r["timings"].items() if "FusedScan" in lbl))
        k_us = sum(us[2:]) / len(us[2:])
        moved, _ = q1.scan_traffic()              # bytes the kernel form that ran moves (counted: vdl_plan_scan_traffic)
        also["tpch_q1_same_rows"] = {"rows": rows, "ms_per_query": 1e3 * sum(wall[2:]) / len(wall[2:]), "rows_per_s": rows / (sum(wall[2:]) / len(wall[2:])),
                                     "kernel_us": k_us, "bytes_per_row": datagen.Q1_BYTES_PER_ROW, "bytes_moved_per_launch": moved,
                                     "roofline_frac": moved / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                     "algorithmic_equivalent_frac": rows * datagen.Q1_BYTES_PER_ROW / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                     "groups": len(r["results"]["tmp101"][".count_order"]), "scan_kernels": q1.jit_note(),
                                     "verified_bit_exact_vs_cpu": q1_matches_sql(r["results"], rows, max(1, min(os.cpu_count() or 1, 64)))}
        q1.close()
        for name in datagen.Q1_COLUMNS:
            eng.drop(name)
    except Exception as exc:                      # noqa: BLE001 -- secondary numbers must never take the headline line down
        also["tpch_q1_same_rows"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    try:
        n_orders = 15000000                       # SF10: 60 M lineitems, 15 M orders, 1.5 M customers
        keep = datagen.register_q3_columns(eng, n_orders)
        q3 = eng.parse(open(os.path.join(ROOT, "tests", "golden", "q3.vdl")).read())
        if jit != "off":
            q3.set_jit(True)                      # (its projection / dimension scans; nothing to tune: their tile shape is fixed)
        times = {}
        for dev_out in (True, False):
            q3.set_device_outputs(dev_out)
            wall = []
            for k in range(7):
                torch.cuda.synchronize(); t0 = time.perf_counter(); q3.execute(); torch.cuda.synchronize(); wall.append(time.perf_counter() - t0)
            times[dev_out] = sum(wall[2:]) / len(wall[2:])
        res = q3.collect(as_numpy=True)["results"]
        also["tpch_q3_sf10"] = {"lineitem_rows": 4 * n_orders, "ms_per_query": 1e3 * times[False], "lineitem_rows_per_s": 4 * n_orders / times[False],
                                "ms_per_query_results_left_in_hbm": 1e3 * times[True], "result_rows": int(len(res["tmp110"][".revenue"])),
                                "path": "fused front (one projection scan of lineitem: ship-date filter + join filter through the orders bitmap, "
                                        "survivors' columns packed) then Partition / Scatter / Fold statement by statement",
                                "scan_kernels": q3.jit_note(), "verified_vs_numpy_sql": q3_matches_sql(eng, res)}
        q3.close()
        for name in list(keep) + ["customer.c_mktsegment", "orders.o_orderdate", "orders.o_shippriority", "orders.orders_customer",
                                  "lineitem.l_shipdate", "lineitem.l_extendedprice", "lineitem.l_discount"]:
            try:
                eng.drop(name)
            except Exception:                     # noqa: BLE001
                pass
        del keep
    except Exception as exc:                      # noqa: BLE001
        also["tpch_q3_sf10"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    try:
        n_li = datagen.LINEITEM_ROWS["sf10"]
        keep = datagen.register_q14_columns(eng, n_li)
        q14 = eng.parse(open(os.path.join(ROOT, "tests", "golden", "q14.vdl")).read())
        q14.set_profiling(True)
        if jit != "off":
            q14.set_jit(True, tune=jit == "tune")
        us, wall = [], []
        for k in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = q14.run(); wall.append(time.perf_counter() - t0)
            us.append(next(v for lbl, v in r["timings"].items() if "FusedScan" in lbl))
        k_us = sum(us[2:]) / len(us[2:])
        t = datagen.q14_tables(n_li)
        import numpy as np
        sel = (t["lineitem.l_shipdate"] >= 728902) & (t["lineitem.l_shipdate"] <= 728931)              # 1995-09-01 <= d < 1995-10-01
        rev = t["lineitem.l_extendedprice"][sel] * (100 - t["lineitem.l_discount"][sel])
        heap = bytes(t["part.p_type.heap"].tobytes())
        promo_off = {o for o in set(t["part.p_type"].tolist()) if heap[o:o + 5] == b"PROMO"}
        promo = np.isin(t["part.p_type"][t["lineitem.lineitem_part"][sel]], sorted(promo_off))
        want = int(rev[promo].sum()) * 10000 // int(rev.sum())
        moved, _ = q14.scan_traffic()             # lineitem columns only: the part side (p_type, the LIKE table) stays in the caches
        also["tpch_q14_sf10"] = {"lineitem_rows": n_li, "ms_per_query": 1e3 * sum(wall[2:]) / len(wall[2:]), "kernel_us": k_us,
                                 "bytes_per_row": datagen.Q14_BYTES_PER_ROW, "bytes_moved_per_launch": moved,
                                 "roofline_frac": moved / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                 "algorithmic_equivalent_frac": n_li * datagen.Q14_BYTES_PER_ROW / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                 "path": "fused JOIN scan: lineitem columns + part.p_type looked up through the join index + a LIKE table, one pass",
                                 "scan_kernels": q14.jit_note(),
                                 "verified_vs_numpy_sql": r["results"]["tmp65"][".promo_revenue"] == [want]}
        q14.close()
        del keep
    except Exception as exc:                      # noqa: BLE001
        also["tpch_q14_sf10"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if rows >= datagen.LINEITEM_ROWS["sf100"]:
        # BASELINE.json config 5's table sizes on ONE GPU (600 M lineitems, 150 M orders, 15 M customers: it fits), program compiled
        # for the SF100 bounds; `bench.py --query q3 --gpus N` is the same query with the Partition exchange over RCCL
        try:
            n_orders = Q3_ORDERS["sf100"]
            r = q3_measure(eng, "cuda:%d" % torch.cuda.current_device(), n_orders, (0, 4 * n_orders), 1, 5, 2, jit)
            also["tpch_q3_sf100"] = {"lineitem_rows": 4 * n_orders, "ms_per_query_results_left_in_hbm": 1e3 * r["elapsed"] / 5,
                                     "lineitem_rows_per_s": 4 * n_orders / (r["elapsed"] / 5), "ms_per_query": r["host_ms"], "result_rows": r["got"][0],
                                     "scan_kernels": r["note"], "verified_vs_torch_sql_checksums": r["got"] == r["want"]}
            r["plan"].close()
        except Exception as exc:                  # noqa: BLE001
            also["tpch_q3_sf100"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    return also


def launch_ranks(n):
    """`python bench.py --gpus N` started bare (no torchrun environment): start the N ranks ourselves, one process per
    GPU, from this parent -- which has not touched HIP -- and hand their exit code back.  Same command line the driver
    uses: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def dry_run(args, world, rank):
    """--dry-run: the launcher, the rendezvous, the row-range sharding and the merge of the partial words, WITHOUT a GPU
    (CPU boxes, tests/test_bench_launch.py): every rank evaluates Q6 over its shard of a small table with the CPU
    checker, the partials are all-reduced over gloo, and the merged answer must equal the unsharded one.  The line it
    prints carries "dry_run": true and no throughput: it is a plumbing check, never a measurement."""
    import torch
    import torch.distributed as dist

    import mplan2vdl_amd as m
    import oracle
    from mplan2vdl_amd import datagen

    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    total_rows = args.rows or datagen.LINEITEM_ROWS["sf0.01"]
    lo, hi = m.shard_rows(total_rows, rank, world)
    specs = [(datagen.SEED, datagen.col_id(c), datagen.LINEITEM[c].lo, datagen.LINEITEM[c].hi,
              datagen.LINEITEM[c].mul, datagen.LINEITEM[c].add) for c in datagen.Q6_COLUMNS]
    rev, cnt = oracle.sql_q6_generated(specs, lo, hi - lo, threads=1)
    words = torch.tensor([cnt, rev], dtype=torch.int64)
    if world > 1:
        m.merge_partials(words, [m._lib.REDUCE_SUM, m._lib.REDUCE_SUM], dist)
    whole = oracle.sql_q6_generated(specs, 0, total_rows, threads=1)
    ok = (int(words[1]), int(words[0])) == whole
    n_ranks = dist.get_world_size() if world > 1 else 1
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU): launcher + sharding + merge only", "value": None, "unit": "rows/s", "n_gpus": n_ranks,
                          "steps": 0, "warmup": 0, "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                          "dtype": "int64", "data": "synthetic", "dry_run": True, "config": {"workload": "tpch_q6_rows%d" % total_rows},
                          "revenue": int(words[1]), "verified_bit_exact_vs_cpu": ok}))
    return 0 if ok else 1


def main_q3(args):
    """--query q3: BASELINE.json config 5 (Q3: lineitem joined to orders and customer through join indices, sparse-domain GROUP BY).
    N = 1: fused front (dimension scans, projection scan of lineitem) + the group-by tail, whole table on one GPU.  N > 1: lineitem
    sharded by rows, orders co-partitioned, customer replicated; the surviving rows travel by key range in ONE grouped RCCL
    send/receive inside libvdl (vdl_run_sharded) -- the all-to-all of the north star -- and every rank ends with the groups of its
    key range.  value = lineitem rows per second over all GPUs with the four result columns left in HBM."""
    import torch
    import torch.distributed as dist

    import mplan2vdl_amd as m

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if os.environ.get("VDL_BENCH_SHARE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    comm_mode = os.environ.get("VDL_BENCH_COMM", "native")
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    n_orders = (args.rows // 4) if args.rows else Q3_ORDERS[args.sf]
    n_li = 4 * n_orders
    lo, hi = m.shard_rows(n_li, rank, world)
    lo, hi = lo - lo % 4, (hi - hi % 4 if rank < world - 1 else n_li)          # an order's lineitems stay on one rank
    eng = m.Engine(device=local_rank)
    eng.use_torch_stream()
    transport = "none (single GPU)"
    if world > 1:
        if comm_mode == "host":                   # rehearsal on a box with fewer GPUs than ranks: libvdl's host transport over gloo
            def gloo_all_gather(send):
                t = torch.frombuffer(bytearray(send), dtype=torch.uint8)
                parts = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(parts, t)
                return [bytes(x.numpy()) for x in parts]

            def gloo_all_to_all(pieces):
                box = [None] * world
                dist.all_gather_object(box, pieces)
                return [box[src][rank] for src in range(world)]

            eng.comm_init_host(rank, world, gloo_all_gather, gloo_all_to_all)
        else:
            box = [eng.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            eng.comm_init_rccl(rank, world, box[0])
        _, n_ranks, name = eng.comm_info()
        if n_ranks != args.gpus:
            print("bench.py: the communicator has %d rank(s), --gpus asked for %d" % (n_ranks, args.gpus), file=sys.stderr)
            sys.exit(2)
        transport = {"rccl": "RCCL inside libvdl: one count all-gather + ONE ncclGroupStart..End of sends/receives (rows by key range) per query",
                     "host": "libvdl host transport over gloo (REHEARSAL)"}[name]
    r = q3_measure(eng, dev, n_orders, (lo, hi), world, args.steps, args.warmup, args.jit, dist if world > 1 else None, sharded=world > 1)
    elapsed = r["elapsed"]
    sums = torch.tensor([r["got"], r["want"]], dtype=torch.int64)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    verified = bool((sums[0] == sums[1]).all())
    if rank == 0:
        out = {"metric": "lineitem rows/s, TPC-H Q3 %s (join-index gathers + sparse GROUP BY; N > 1: RCCL all-to-all Partition)" % (args.sf.upper() if not args.rows else "%d lineitems" % n_li),
               "value": n_li / (elapsed / args.steps), "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
               "config": {"workload": "tpch_q3_%s" % (args.sf if not args.rows else "rows%d" % n_li), "lineitem_rows": n_li, "orders_rows": n_orders,
                          "customer_rows": max(n_orders // 10, 1), "sharding": "lineitem by rows, orders co-partitioned, customer replicated" if world > 1 else "single GPU",
                          "exchange": transport, "results": "left in HBM (4 columns)"},
               "roofline": None,                  # random access through join indices: rows/s only
               "cpu_baseline": None,
               "result_rows_all_ranks": int(sums[0][0]), "ms_per_query_results_copied_to_host": r["host_ms"],
               "verified_vs_torch_sql_checksums": verified,
               "checksums": {"groups, sum(revenue), sum(orderkey), sum(orderdate), sum(orderkey*revenue) mod 2^64": [int(x) for x in sums[0]]},
               "scan_kernels": r["note"]}
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not verified:
        print("VERIFICATION FAILED (q3): engine %s vs torch %s" % (sums[0].tolist(), sums[1].tolist()), file=sys.stderr)
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=0, help="total lineitem rows (default: SF100 = 600,037,902)")
    ap.add_argument("--sf", type=str, default="sf100", help="sf0.01 | sf1 | sf10 | sf100")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=59986052, help="rows of the CPU-baseline sample (default SF10)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the Q1 / Q3 numbers reported under \"also\" (N=1 only)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--query", default="q6", choices=["q6", "q1", "q3"],
                    help="q6 (default, BASELINE.json's metric), q1 (grouped fused scan; secondary measurement) or q3 (join + sparse GROUP BY; "
                         "N > 1: the Partition exchange over RCCL -- BASELINE.json config 5)")
    ap.add_argument("--jit", default="tune", choices=["off", "on", "tune"],
                    help="scan kernels specialised for the plan by hiprtc at the first (untimed) run: off = the precompiled kernels, on = specialised with the "
                         "precompiled variant's rows per lane, tune (default) = rows per lane chosen by timing at that run, the precompiled kernel staying if it wins")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: check launcher / sharding / merge plumbing only (see dry_run)")
    ap.add_argument("--latency-steps", type=int, default=10, help="queries run one at a time after the timed region to report per-query latency")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))         # nothing below runs in the parent: it never touches HIP
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print("bench.py: --gpus %d but the launcher started %s rank(s)" % (args.gpus, os.environ.get("WORLD_SIZE", "1")), file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        sys.exit(dry_run(args, int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))))
    if args.query == "q3":
        if args.jit != "off":
            base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
            os.environ.setdefault("VDL_JIT_CACHE", os.path.join(base, "vdl-mi355x", "jit"))
        return main_q3(args)

    import torch
    import torch.distributed as dist

    import mplan2vdl_amd as m
    from mplan2vdl_amd import datagen

    if args.jit != "off":
        # code objects of the specialised scans are kept across processes (ranks of one run, the N = 1, 2, 4, 8 runs of a
        # scaling sweep): the key is a hash of the whole generated source, so a changed kernel or plan never meets a stale one.
        # The directory is private to this user (libvdl creates it 0700 and refuses one that anybody else could write:
        # csrc/vdl_jit.cpp "the on-disk cache"); no usable home directory = no cache, every process compiles for itself.
        base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
        os.environ.setdefault("VDL_JIT_CACHE", os.path.join(base, "vdl-mi355x", "jit"))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    # torch.distributed is the launcher's control channel only (communicator id hand-over, barriers, MAX of the timings):
    # gloo over CPU tensors.  The data-path collectives run inside libvdl on its own RCCL communicator (vdl_comm_init).
    # Rehearsal on one-GPU boxes: VDL_BENCH_SHARE_DEVICE=1 puts every rank on device 0; RCCL refuses two ranks on one
    # device, so VDL_BENCH_COMM=host then routes the collectives through libvdl's host transport over gloo: a rehearsal, not for
    # reported numbers; the JSON line says which one ran.  When libvdl cannot bind RCCL the run fails (exit code 3).
    comm_mode = os.environ.get("VDL_BENCH_COMM", "native")
    if os.environ.get("VDL_BENCH_BACKEND") == "gloo" and comm_mode == "native":
        comm_mode = "host"                                        # (round-1 spelling of the rehearsal switch)
    if os.environ.get("VDL_BENCH_SHARE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            print("bench.py: the launcher's group has %d rank(s), --gpus asked for %d" % (dist.get_world_size(), args.gpus), file=sys.stderr)
            sys.exit(2)

    def everybody(ok):
        """True when `ok` holds on every rank (one gloo all-reduce)."""
        if world == 1:
            return bool(ok)
        t = torch.tensor([0 if ok else 1], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item()) == 0

    total_rows = args.rows or datagen.LINEITEM_ROWS[args.sf]
    lo, hi = m.shard_rows(total_rows, rank, world)
    my_rows = hi - lo

    eng = m.Engine(device=local_rank)
    q_cols = datagen.Q6_COLUMNS if args.query == "q6" else datagen.Q1_COLUMNS
    q_bytes = datagen.Q6_BYTES_PER_ROW if args.query == "q6" else datagen.Q1_BYTES_PER_ROW
    for name in q_cols:
        eng.generate(datagen.LINEITEM[name], lo, my_rows)
    text = open(os.path.join(ROOT, "tests", "golden", args.query + ".vdl")).read()
    plan = eng.parse(text)
    if not plan.is_fused:
        raise SystemExit("%s did not fuse:\n%s" % (args.query, plan.describe()))
    plan.set_profiling(True)
    plan.set_row_offset(lo)
    if args.jit != "off":
        plan.set_jit(True, tune=args.jit == "tune")
    nw, ops = plan.partial_spec()

    n_ranks, transport = 1, "none (single GPU)"
    if world > 1 and comm_mode == "native":
        # every rank binds RCCL first (so that nobody blocks in ncclCommInitRank while a peer could not even load the
        # library), rank 0's id travels over the control channel, then the communicator is built
        uid, why = None, ""
        try:
            uid = eng.comm_unique_id()
        except m.VdlError as exc:
            why = str(exc)
        if everybody(uid is not None):
            box = [uid if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            try:
                eng.comm_init_rccl(rank, world, box[0])
                why = ""
            except m.VdlError as exc:
                why = str(exc)
        if not everybody(uid is not None and why == ""):
            # no quiet fallback: the collectives of a reported number run inside libvdl over RCCL, or the run fails (round 4: the
            # torch.distributed all-reduce formulation from Python is gone from this script)
            print("bench.py[%d]: libvdl could not set up RCCL%s" % (rank, (": " + why) if why else " (another rank failed)"), file=sys.stderr)
            sys.exit(3)
    if world > 1 and comm_mode == "host":
        def gloo_all_gather(send):
            t = torch.frombuffer(bytearray(send), dtype=torch.uint8)
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            return [bytes(x.numpy()) for x in parts]

        def gloo_all_to_all(pieces):                              # (the fold route never exchanges rows; complete for the interface)
            box = [None] * world
            dist.all_gather_object(box, pieces)
            return [box[src][rank] for src in range(world)]

        eng.comm_init_host(rank, world, gloo_all_gather, gloo_all_to_all)
    if world > 1 and comm_mode in ("native", "host"):
        _, n_ranks, name = eng.comm_info()
        transport = {"rccl": "RCCL inside libvdl: one all-gather of %d int64 words (partial words twice + status) + merge kernel per query, on a communication stream",
                     "host": "libvdl host transport over gloo (REHEARSAL): one all-gather of %d int64 words + merge kernel per query"}[name] % (2 * nw + 1)
        if n_ranks != args.gpus:
            print("bench.py: the communicator has %d rank(s), --gpus asked for %d" % (n_ranks, args.gpus), file=sys.stderr)
            sys.exit(2)
    query, bufs = None, None
    if world == 1:
        # one rank: local phase -> finalisation, pipelined over two partial buffers (no communicator, nothing to merge)
        side = torch.cuda.Stream()
        torch.cuda.set_stream(side)
        eng.use_torch_stream()
        bufs = [torch.zeros(max(nw, 1), dtype=torch.int64, device="cuda") for _ in range(2)]
        query = m.ShardedQuery(plan, bufs[0])

    scan_us = []

    def run_steps(k, record):
        """k full queries, pipelined: the host side of query i (and, N > 1, its collective) overlaps the scan of query
        i+1; every query runs completely and every result is produced."""
        def on_result(out):
            if record:
                scan_us.append(plan.scan_stats()[2])
        if query is not None:
            return query.run_pipelined(k, bufs, on_result, overlap_merge=os.environ.get("VDL_BENCH_SYNC_MERGE") != "1")
        out = None
        for i in range(k):
            plan.run_sharded_begin(i & 1)
            if i >= 1:
                out = plan.run_sharded_end(1 - (i & 1))
                on_result(out)
        if k:
            out = plan.run_sharded_end((k - 1) & 1)
            on_result(out)
        return out

    def one_query():
        return query.step() if query is not None else plan.run_sharded()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1 or args.jit != "off":
        # communicator warm-up (the first collective takes seconds) and the build / tuning of the specialised scan kernels
        # (hiprtc: seconds) stay out of the timed region even with --warmup 0
        one_query()
        torch.cuda.synchronize()
    result = run_steps(args.warmup, False) if args.warmup > 0 else None
    sync_all()
    t0 = time.perf_counter()
    result = run_steps(args.steps, True)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-query latency: the same query run one at a time (local phase, merge, finalisation, answer on the host before
    # the next one starts) -- `value` above is pipelined throughput, this is what a single request waits for
    latency_ms = None
    if args.latency_steps > 0:
        lat = []
        for _ in range(args.latency_steps):
            sync_all()
            t1 = time.perf_counter()
            one_query()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t1)
        lt = torch.tensor([sum(lat[1:]) / max(len(lat) - 1, 1) if len(lat) > 1 else lat[0]], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(lt, op=dist.ReduceOp.MAX)
        latency_ms = float(lt.item()) * 1e3

    revenue = result["results"]["tmp42"][".revenue"] if args.query == "q6" else None
    kernel_label = next((k.replace("timeInMicrosecondsForFusedScan_", "") for k in result["timings"] if "FusedScan" in k), "k_scan")
    ms_per_step = elapsed / args.steps * 1e3
    rows_per_s = total_rows / (elapsed / args.steps)
    kern_us = sum(scan_us) / len(scan_us)

    out = None
    if rank == 0:
        verified = None
        cpu_baseline = None
        if not args.no_verify or not args.no_cpu_baseline:
            import oracle
        # torchrun exports OMP_NUM_THREADS=1; the checker may use the host's cores (capped)
        host_threads = max(1, min(os.cpu_count() or 1, 64))
        if not args.no_verify and args.query == "q1":
            verified = q1_matches_sql(result["results"], total_rows, host_threads, lambda msg: print(msg, file=sys.stderr))
            if not verified:
                print("VERIFICATION FAILED (q1)", file=sys.stderr)
        if not args.no_verify and args.query == "q6":
            # bit-exact check of the full-size answer: the SQL-semantics loop over regenerated rows
            # on all host cores (test infrastructure; outside the timed region)
            specs = [(datagen.SEED, datagen.col_id(c), datagen.LINEITEM[c].lo, datagen.LINEITEM[c].hi,
                      datagen.LINEITEM[c].mul, datagen.LINEITEM[c].add) for c in datagen.Q6_COLUMNS]
            rev, cnt = oracle.sql_q6_generated(specs, 0, total_rows, threads=host_threads)
            verified = (revenue == ([rev] if cnt else []))
            if not verified:
                print("VERIFICATION FAILED: gpu %r vs cpu %r" % (revenue, rev), file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline and args.query == "q6":
            # CPU baseline = the scalar op-at-a-time oracle interpreter on a bounded sample of the
            # same workload (first SF10 rows), one host core
            n_s = min(args.cpu_sample_rows, total_rows)
            orc = oracle.Oracle()
            for name in datagen.Q6_COLUMNS:
                orc.add_column(name, datagen.generate(datagen.LINEITEM[name], 0, n_s))
            r = orc.run(text)
            secs = orc.last_seconds
            cols = [datagen.generate(datagen.LINEITEM[c], 0, n_s) for c in datagen.Q6_COLUMNS]
            t1 = time.perf_counter(); rev1, cnt1 = oracle.sql_q6(*cols, threads=1); f1 = time.perf_counter() - t1
            nt = host_threads
            t1 = time.perf_counter(); oracle.sql_q6(*cols, threads=nt); fn = time.perf_counter() - t1
            ok = r["results"]["tmp42"][".revenue"] == ([rev1] if cnt1 else [])
            cpu_baseline = {"value": n_s / secs, "unit": "rows/s", "cores": 1, "kind": "port",
                            "sample": "first %d rows of the same synthetic lineitem, Q6 VDL, scalar op-at-a-time interpreter, %.2f s" % (n_s, secs),
                            "fused_sql_loop_rows_per_s_1core": n_s / f1,
                            "fused_sql_loop_rows_per_s_allcores": n_s / fn, "allcores": nt,
                            "interpreter_matches_sql_loop": ok}
        # ---- bytes the timed kernel MOVED (roofline.achieved is built on them, so frac <= 1 by construction) ------------------------
        # libvdl counts them for the kernel form that ran (vdl_plan_scan_traffic): a scan that reads everything moves its algorithmic
        # bytes; a staged scan (late materialisation) moves the eager columns plus 128 B per cache line of a late column that still
        # held a live row -- counted by a census build of the same form in one untimed launch.  Whole 128-byte lines are what the
        # memory side fetches: tools/ubench/fetch_calib.hip, profiles/r03/fetch_calib.txt.  `traffic` is the independent figure:
        # FETCH_SIZE x 2 of the committed rocprofv3 --pmc pass, used only when it was taken for EXACTLY this kernel (name with
        # rows per lane, stages, grid) over the same number of rows (tools/profile_bench.sh -> profiles/traffic.json).
        moved, moved_detail = plan.scan_traffic()
        achieved = moved / (kern_us * 1e-6) / 1e9 if kern_us > 0 else 0.0
        tuner_note = plan.jit_note()                  # (before the read-everything run below switches the specialisation off)
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath)).get(kernel_label, {})
                if tj.get("rows") == my_rows and tj.get("query") == args.query:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = "committed profile of this kernel, not measured in this run: " + tj.get("source", "profiles/traffic.json")
            except Exception:
                traffic = None
        # the kernel that reads every byte (precompiled k_scan / k_mscan), timed in this same run after the timed region: its
        # fraction is the plain "algorithmic bytes / time" figure of SURVEY.md 8(d)
        read_everything = None
        if world == 1 and args.jit != "off" and not args.no_secondary:
            try:
                plan.set_jit(False)
                us = []
                for _ in range(12):
                    plan.run()
                    us.append(plan.scan_stats()[2])
                r_label = next((k.replace("timeInMicrosecondsForFusedScan_", "") for k in plan.run()["timings"] if "FusedScan" in k), "?")
                k_us = sum(us[2:]) / len(us[2:])
                read_everything = {"kernel": r_label, "kernel_us": k_us, "bytes_per_launch": my_rows * q_bytes,
                                   "achieved": my_rows * q_bytes / (k_us * 1e-6) / 1e9, "frac": my_rows * q_bytes / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBPS}
            except Exception as exc:              # noqa: BLE001
                read_everything = {"error": "%s: %s" % (type(exc).__name__, exc)}
        out = {
            "metric": "rows/s, TPC-H %s %s (fused VDL scan), + achieved HBM GB/s in roofline" % (args.query.upper(), args.sf.upper() if not args.rows else "%d rows" % total_rows),
            "value": rows_per_s, "unit": "rows/s", "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "latency_ms_per_query": latency_ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic",
            "config": {"workload": "tpch_%s_%s" % (args.query, args.sf if not args.rows else "rows%d" % args.rows),
                       "rows_total": total_rows, "rows_per_gpu": my_rows, "bytes_per_row": q_bytes,
                       "sharding": "row-range, one process per GPU" if world > 1 else "single GPU",
                       "finalise": transport if world > 1 else "local"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         # (the guide that names the 8 TB/s peak also measures what a plain copy reaches on this part: the practical ceiling)
                         "achievable_per_guide": HBM_ACHIEVABLE_GBPS, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBPS,
                         # SURVEY.md 8(d)'s own figure -- algorithmic bytes (every column of every row, once) / time / peak -- of the kernel
                         # that really reads them all, timed in this run after the timed region (null with --jit off --no-secondary etc.)
                         "frac_algorithmic_read_everything": (read_everything or {}).get("frac"),
                         "definition": {"frac": "bytes the TIMED kernel moved (counted per launch: bytes_moved_per_launch; = FETCH_SIZE x 2 of its committed "
                                                "profile, traffic) / its mean launch time / peak: <= 1 by construction; a staged scan skips cache lines "
                                                "that hold no live row, so this is below the algorithmic 28 B/row rate",
                                        "frac_algorithmic_read_everything": "SURVEY.md 8(d): rows x bytes_per_row / launch time / peak for the kernel "
                                                                            "that reads every column of every row (read_everything_kernel), same run",
                                        "algorithmic_equivalent_GBps": "rows x bytes_per_row / the TIMED kernel's launch time: may exceed the peak, it "
                                                                       "prices bytes the kernel did not move"},
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_label, "kernel_us": kern_us,
                         "bytes_moved_per_launch": moved,
                         "bytes_moved_source": "vdl_plan_scan_traffic: " + moved_detail,
                         "algorithmic_bytes_per_launch": my_rows * q_bytes,
                         # what the same time means in SURVEY.md 8(d)'s terms (every column of every row counted once): above the
                         # HBM peak when the kernel reads late -- it evaluates the same rows, it does not move those bytes
                         "algorithmic_equivalent_GBps": my_rows * q_bytes / (kern_us * 1e-6) / 1e9 if kern_us > 0 else 0.0,
                         "late_materialisation": moved < my_rows * q_bytes,
                         "read_everything_kernel": read_everything},
            "cpu_baseline": cpu_baseline,
            "revenue": (revenue[0] if revenue else None), "verified_bit_exact_vs_cpu": verified,
            "scan_kernels": {"mode": args.jit, "note": tuner_note},
        }
        if world == 1 and args.query == "q6" and not args.no_secondary:
            out["also"] = secondary_measurements(eng, total_rows, args.jit)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))
        if out["verified_bit_exact_vs_cpu"] is False:
            sys.exit(1)


if __name__ == "__main__":
    main()

#!/bin/bash
# rocprofv3 evidence for bench.py's roofline objects: kernel-trace stats of the headline run (Q6 SF100) and of Q1 over the
# same rows, plus separate --pmc FETCH_SIZE / WRITE_SIZE passes (never combined with other trace domains).  Leaves
#   gpurun_out/profile_bench/{q6,q1}_kernel_stats.csv, {q6,q1}_bench.json, traffic.json
# which are copied under profiles/rNN/ (and traffic.json to profiles/) by hand afterwards.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/profile_bench
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (deterministic kernels for the profiles: Q6 on the hand-tuned k_scan -- the tuner's choice for it -- and Q1 on the kernel
# specialised for the plan at 6 row pairs per lane -- the tuner's choice for it -- without the tuner's trial launches)
export VDL_JIT_GROUP_U=6
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q6 -- python3 $ROOT/bench.py --jit off --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/q6_bench.json 2> $OUT/q6.err
# Q6 on the kernel the tuner picks on the boxes tried last: specialised, 2 row pairs per lane, staged reads (ship date with the
# tile, then discount, quantity and extended price for the rows still in; with two columns in the tile it is within 3 %)
VDL_JIT_U=2 VDL_JIT_LATE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q6late -- python3 $ROOT/bench.py --jit on --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/q6late_bench.json 2> $OUT/q6late.err
VDL_JIT_U=2 VDL_JIT_LATE=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/q6late_fetch -- python3 $ROOT/bench.py --jit on --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --latency-steps 0 > $OUT/q6late_fetch.json 2> $OUT/q6late_fetch.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q1 -- python3 $ROOT/bench.py --jit on --query q1 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/q1_bench.json 2> $OUT/q1.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/q6_fetch -- python3 $ROOT/bench.py --jit off --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --latency-steps 0 > $OUT/q6_fetch.json 2> $OUT/q6_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/q6_write -- python3 $ROOT/bench.py --jit off --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --latency-steps 0 > $OUT/q6_write.json 2> $OUT/q6_write.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/q1_fetch -- python3 $ROOT/bench.py --jit on --query q1 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --latency-steps 0 > $OUT/q1_fetch.json 2> $OUT/q1_fetch.err
# the plain command (default --jit tune, secondary measurements included): the tuner's candidates show up as kernels of their
# own (vdl_jit_mscan_u<pairs>[_staged]), the winner's average is the one bench.py reports
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/default_bench.json 2> $OUT/default.err
cp $(ls $OUT/default/*/*kernel_stats.csv | head -n 1) $OUT/default_kernel_stats.csv
for q in q6 q6late q1; do
  f=$(ls $OUT/$q/*/*kernel_stats.csv | head -n 1); cp $f $OUT/${q}_kernel_stats.csv
  echo "== $q"; head -n 6 $f | cut -c1-160; tail -n 1 $OUT/${q}_bench.json | cut -c1-300
done
python3 - $OUT <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
def per_dispatch(tag, needle):
    f = glob.glob(out + "/%s/*/*counter_collection.csv" % tag)[0]
    tot, ids, name = 0.0, set(), None
    for r in csv.DictReader(open(f)):
        if needle in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); ids.add(r["Dispatch_Id"]); name = r["Kernel_Name"]
    return tot / max(len(ids), 1), len(ids), name
# unit calibration on the generator: WRITE_SIZE of k_gen_column<long> against the bytes it is known to write (KiB per count?)
res = {}
for q, needle, rows_key in (("q6", "k_scan<", "q6_fetch"), ("q6_late", "vdl_jit_mscan", "q6late_fetch"), ("q1", "vdl_jit_mscan", "q1_fetch")):
    fetch, n, name = per_dispatch(rows_key, needle)
    bench = json.loads(open(out + "/%s.json" % rows_key).read().strip().splitlines()[-1])
    rows = bench["config"]["rows_per_gpu"]
    algo = bench["roofline"]["algorithmic_bytes_per_launch"]
    res[q] = {"rows": rows, "kernel": name, "FETCH_SIZE_KiB_mean": fetch, "dispatches": n,
              "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced streaming reads -> x2 (MI355X_MICROARCH.md, HBM section); unit = KiB",
              "hbm_bytes_per_launch": int(fetch * 1024 * 2), "algorithmic_bytes_per_launch": algo,
              "source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py %s--steps 3 --warmup 1 (tools/profile_bench.sh)" % ({"q1": "--jit on --query q1 ", "q6": "--jit off ", "q6_late": "--jit on [VDL_JIT_U=2 VDL_JIT_LATE=1] "}[q])}
    print(q, name, "FETCH_SIZE/dispatch %.0f KiB -> %.4g B (x2), algorithmic %.4g B, ratio %.5f" % (fetch, fetch * 2048, algo, fetch * 2048 / algo))
w, n, name = per_dispatch("q6_write", "k_scan<")
print("q6 WRITE_SIZE/dispatch %.1f KiB over %d dispatches" % (w, n))
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
PY

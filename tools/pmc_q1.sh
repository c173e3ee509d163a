#!/bin/bash
# PMC passes over the grouped fused scan (bench.py --query q1); run on the GPU box from the repo root.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_q1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--query q1 --sf ${1:-sf10} --steps 3 --warmup 1 --no-cpu-baseline --no-verify"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU \
  --kernel-trace --output-format csv -d $OUT/a -- python3 $ROOT/bench.py $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_SMEM \
  --kernel-trace --output-format csv -d $OUT/b -- python3 $ROOT/bench.py $ARGS > $OUT/b.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for d in ("a", "b"):
    for f in glob.glob(root + "/gpurun_out/pmc_q1/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "mscan<" in k or "k_scan<" in k:
                print(d, k, {c: "%.3g" % x for c, x in v.items()})
PY

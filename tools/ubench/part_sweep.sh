#!/bin/bash
# builds tools/ubench/part_bench.hip in several tile shapes (here, on the CPU box) or runs what was built (on the GPU box)
#   tools/ubench/part_sweep.sh build    |    tools/ubench/part_sweep.sh run <tag>
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=$ROOT/tools/ubench/part_bench_bin
VARIANTS=("512 16 4" "512 16 4 -DVDL_PART_NOTICKET")
if [ "$1" = build ]; then
    mkdir -p $B; rm -f $B/*
    i=0
    for v in "${VARIANTS[@]}"; do
        set -- $v
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$ROOT/include -I$ROOT/mplan2vdl_amd/csrc -DVDL_PART_BLOCK=$1 -DVDL_PART_STEPS=$2 -DVDL_PART_EU=$3 ${@:4} $ROOT/tools/ubench/part_bench.hip -o $B/v$i &
        i=$((i+1))
    done
    wait; ls -la $B
else
    OUT=$ROOT/gpurun_out/${2:-part_sweep}; mkdir -p $OUT
    cd /tmp && export TMPDIR=/tmp
    for f in $B/v*; do
        rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$(basename $f) -- $f > $OUT/$(basename $f).log 2>&1
        tail -n 1 $OUT/$(basename $f).log
        python3 - $OUT/$(basename $f) <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:8]:
        if "k_part" in r["Name"] or "fill" in r["Name"]: print("    %-84s %4s x %8.1f us" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    done
fi

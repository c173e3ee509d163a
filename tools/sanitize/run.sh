#!/bin/bash
# AddressSanitizer + UBSan over (a) the oracle (C) running the fixtures, (b) the engine's CPU-only front
# door (parser + fusion planner).  GPU ASan is not available on this pool.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
OUT=/tmp/vdl_sanitize
mkdir -p $OUT
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -I$HERE/include -I$HERE/mplan2vdl_amd/csrc \
    $HERE/tools/sanitize/parse_fuse_main.cpp $HERE/mplan2vdl_amd/csrc/vdl_parse.cpp $HERE/mplan2vdl_amd/csrc/vdl_fuse.cpp -o $OUT/parse_fuse
# every plan the front end compiles, in both dialects, goes through the sanitized parser + planner too
PYTHONPATH=$HERE python3 - $HERE $OUT <<'PY'
import os, sys
from mplan2vdl_amd import frontend
here, out = sys.argv[1:3]
meta = os.path.join(here, "tests", "golden", "tpch10noorder")
for fmt in ("vdl", "vlite"):
    cfg = frontend.load_metadata(meta, format=fmt)
    for n in (1, 3, 4, 5, 6, 9, 10, 11, 12, 14, 15, 16, 18, 19, 20):
        open(os.path.join(out, "q%02d.%s" % (n, fmt)), "w").write(frontend.compile_plan(open(os.path.join(meta, "%02d.sql.mplan" % n)).read(), cfg))
PY
# ... and the random join / condition / front programs of the parity tests (round 2: condition columns, column ordering, dimension scans)
mkdir -p $OUT/rand
PYTHONPATH=$HERE:$HERE/tests python3 - $OUT/rand <<'PY'
import sys
from test_random_conditions import Gen, FrontGen
from test_random_joins import Gen as JoinGen
from test_random_fused import Gen as FusedGen
from test_random_semijoins import Gen as SemiGen
k = 0
for G in (Gen, FrontGen, JoinGen, FusedGen, SemiGen):
    for seed in range(120):
        open("%s/p%04d.vdl" % (sys.argv[1], k), "w").write(G(seed).build()[0]); k += 1
PY
$OUT/parse_fuse $HERE/tests/golden/q6.vdl $HERE/tests/golden/q1.vdl $HERE/tests/golden/q3.vdl $OUT/q*.vdl $OUT/q*.vlite $OUT/rand/*.vdl
gcc -std=c11 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp -shared -fPIC $HERE/oracle/vdl_oracle.c -o $OUT/libvdl_oracle_asan.so
echo "oracle ASan build ok: $OUT/libvdl_oracle_asan.so"
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 VDL_ORACLE_SO=$OUT/libvdl_oracle_asan.so \
    python3 -m pytest $HERE/tests/test_oracle.py $HERE/tests/test_datagen.py $HERE/tests/test_tpch_plans.py -m "not gpu" -x -q -p no:cacheprovider

#!/bin/bash
# AddressSanitizer + UBSan over (a) the oracle (C) running the fixtures, (b) the engine's CPU-only front
# door (parser + fusion planner).  GPU ASan is not available on this pool.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
OUT=/tmp/vdl_sanitize
mkdir -p $OUT
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -I$HERE/include -I$HERE/mplan2vdl_amd/csrc \
    $HERE/tools/sanitize/parse_fuse_main.cpp $HERE/mplan2vdl_amd/csrc/vdl_parse.cpp $HERE/mplan2vdl_amd/csrc/vdl_fuse.cpp -o $OUT/parse_fuse
$OUT/parse_fuse $HERE/tests/golden/q6.vdl $HERE/tests/golden/q1.vdl $HERE/tests/golden/q3.vdl
gcc -std=c11 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp -shared -fPIC $HERE/oracle/vdl_oracle.c -o $OUT/libvdl_oracle_asan.so
echo "oracle ASan build ok: $OUT/libvdl_oracle_asan.so"
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 VDL_ORACLE_SO=$OUT/libvdl_oracle_asan.so \
    python3 -m pytest $HERE/tests/test_oracle.py $HERE/tests/test_datagen.py -x -q -p no:cacheprovider

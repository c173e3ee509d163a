#!/bin/bash
# A/B of libvdl builds on the grouped scan: tools/q1_ab.sh [sf] lib1 lib2 ...   ("default" = the in-tree build)
SF=${1:-sf10}; shift
for lib in "$@"; do
  if [ "$lib" = default ]; then unset VDL_LIB; else export VDL_LIB=$PWD/mplan2vdl_amd/lib/exp/libvdl_$lib.so; fi
  timeout -k 10 300 python bench.py --query q1 --sf $SF --steps 20 --warmup 3 --no-cpu-baseline $Q1_AB_EXTRA > gpurun_out/q1_ab_$lib.log 2>&1 || echo "$lib: bench exited non-zero (expected for the timing-only ablation builds, whose results are wrong)"
  tail -n 1 gpurun_out/q1_ab_$lib.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['roofline']['kernel'], '%.1f us' % d['roofline']['kernel_us'], 'frac %.3f' % d['roofline']['frac'], 'exact', d.get('verified_bit_exact_vs_cpu'))"
done

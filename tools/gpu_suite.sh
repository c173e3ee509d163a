#!/bin/bash
# GPU test suite with the full log kept (gpurun_out/gpu_suite_<tag>.log); prints the summary line and, on failure, the report
TAG=${1:-default}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd $ROOT
python -m pytest tests -m gpu -q > gpurun_out/gpu_suite_$TAG.log 2>&1
rc=$?
tail -n 1 gpurun_out/gpu_suite_$TAG.log
if [ $rc -ne 0 ]; then grep -n -A80 "^=* FAILURES" gpurun_out/gpu_suite_$TAG.log | cut -c1-260 | head -n 140; fi
exit $rc

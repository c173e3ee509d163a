#!/bin/bash
# Q3 at SF10 with and without the specialised projection / dimension scans (tools/q3_statement_profile.py)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
echo "== precompiled"; python3 $ROOT/tools/q3_statement_profile.py 2>&1 | grep -v amdgpu.ids | head -8
echo "== specialised"; VDL_JIT=1 python3 $ROOT/tools/q3_statement_profile.py 2>&1 | grep -v amdgpu.ids | head -8

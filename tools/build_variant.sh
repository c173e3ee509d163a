#!/bin/bash
# tools/build_variant.sh TAG [-DFLAG=...]: a copy of libvdl.so with vdl_mscan.hip and vdl_jit.cpp rebuilt under extra flags (the scans' precompiled
# kernels, and what the run-time specialisation hands to hiprtc), for A/B runs on the GPU box:  VDL_LIB=mplan2vdl_amd/lib/exp/libvdl_TAG.so python bench.py ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
OUT=$ROOT/mplan2vdl_amd/lib/exp
mkdir -p "$OUT"
cd "$ROOT/mplan2vdl_amd/csrc"
L=$ROOT/mplan2vdl_amd/lib
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -Wall -Wno-unused-result "$@" -c vdl_mscan.hip -o "$OUT/vdl_mscan_$TAG.o" &
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$L" -Wall -Wno-unused-result "$@" -c vdl_jit.cpp -o "$OUT/vdl_jit_$TAG.o" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$OUT/libvdl_$TAG.so" "$L/vdl_parse.o" "$L/vdl_fuse.o" "$L/vdl_kernels.o" "$L/vdl_ops.o" "$L/vdl_partition.o" "$OUT/vdl_mscan_$TAG.o" "$L/vdl_engine.o" "$L/vdl_exchange.o" "$L/vdl_comm.o" "$OUT/vdl_jit_$TAG.o" -ldl
rm -f "$OUT/vdl_mscan_$TAG.o" "$OUT/vdl_jit_$TAG.o"
echo "$OUT/libvdl_$TAG.so"

#!/bin/bash
# Builds a variant of libroft_hip.so with extra compiler flags into build_ab/<name>.so (A/B runs: ROFT_LIB_SO=build_ab/<name>.so).
# usage: bash tools/build_variant.sh <name> "<extra flags>"
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=${ROFT_SRC:-$R}
NAME=$1; EXTRA=$2
B=$R/build_ab/obj_$NAME; rm -rf $B; mkdir -p $B
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-pass-failed $EXTRA"
pids=()
for f in k_mask k_flow k_skf k_ukf k_render k_opticalflow engine engine_submit engine_step engine_results engine_ops engine_debug flow_producer mesh_class; do
  [ -f $SRC/roft_amd/csrc/$f.hip ] || continue   # (a source tree of an earlier commit may lack a file)
  hipcc $FLAGS -c $SRC/roft_amd/csrc/$f.hip -o $B/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -o $R/build_ab/$NAME.so $B/*.o
echo built $R/build_ab/$NAME.so

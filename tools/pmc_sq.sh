#!/bin/bash
# One SQ counter pass (instruction counts and wave time split) over the native driver with the library
# given as $1 put in librtx.so's place (on the GPU box's scratch copy).  usage: tools/pmc_sq.sh <lib.so> <tag>
set -u
LIB=$1; TAG=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cp "$ROOT/$LIB" "$ROOT/ray_tracer_amd/librtx.so" || exit 1
OUT=$ROOT/gpurun_out/pmc/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--scene dragon --standin 871316 --width 3840 --height 2160 --mode pruned --repeats 3 --out /tmp/pmc_canvas.ppm"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d "$OUT/sq1" -- "$ROOT/ray_tracer_amd/rtx_main" $ARGS > "$OUT/sq1.log" 2>&1 || { echo "pass failed"; tail -3 "$OUT/sq1.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_SMEM --output-format csv -d "$OUT/sq2" -- "$ROOT/ray_tracer_amd/rtx_main" $ARGS > "$OUT/sq2.log" 2>&1 || { echo "pass 2 failed"; tail -3 "$OUT/sq2.log"; exit 1; }
echo done

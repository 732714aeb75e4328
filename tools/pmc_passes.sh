#!/bin/bash
# Runs on the GPU box (through gpurun): one rocprofv3 run per counter group over the native
# driver, each under its own time limit; CSVs land in gpurun_out/pmc/<tag>/<pass>/.
# usage: tools/pmc_passes.sh <tag> <scene> <standin-tris|0> <W> <H> <mode> [knot]
set -u
TAG=$1; SCENE=$2; TRIS=$3; W=$4; H=$5; MODE=$6; FAMILY=${7:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "$TRIS" != "0" ]; then SRC="--standin $TRIS"; [ "$FAMILY" = "knot" ] && SRC="$SRC --knot"; else SRC="--obj $ROOT/tests/golden/$SCENE.obj"; fi
ARGS="--scene $SCENE $SRC --width $W --height $H --mode $MODE --repeats 3 --out /tmp/pmc_canvas.ppm"
run() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- "$ROOT/ray_tracer_amd/rtx_main" $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
}
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "$ROOT/ray_tracer_amd/rtx_main" $ARGS > "$OUT/trace.log" 2>&1 || exit 1
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU || exit 1
run sq2 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS || exit 1
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum || exit 1
run grbm GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
echo done

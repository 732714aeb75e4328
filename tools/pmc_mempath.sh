#!/bin/bash
# Counter passes over the vector-memory path of the render kernel (address unit TA, vector L1 TCP, data
# return TD), one rocprofv3 run per group, over the native driver.  Runs on the GPU box (through gpurun);
# CSVs land in gpurun_out/pmc/<tag>/<pass>/ for tools/pmc_summary.py.
# usage: tools/pmc_mempath.sh <tag> <scene> <standin-tris|0> <W> <H> <mode> [spp]
set -u
TAG=$1; SCENE=$2; TRIS=$3; W=$4; H=$5; MODE=$6; SPP=${7:-1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "$TRIS" != "0" ]; then SRC="--standin $TRIS"; else SRC="--obj $ROOT/tests/golden/$SCENE.obj"; fi
ARGS="--scene $SCENE $SRC --width $W --height $H --mode $MODE --spp $SPP --repeats 3 --out /tmp/pmc_canvas.ppm"
run() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- "$ROOT/ray_tracer_amd/rtx_main" $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
}
run busy TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE || exit 1
run stall TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TD_SPI_STALL_sum TD_LOAD_WAVEFRONT_sum || exit 1
run lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum || exit 1
run tag TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum || exit 1
run sq SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES || exit 1
run utcl TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_RFIFO_STALL_CYCLES_sum || exit 1
echo done

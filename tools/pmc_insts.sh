#!/bin/bash
# Instruction counts of the render kernel by class, one rocprofv3 counter pass over the native driver (GPU box).
# usage: tools/pmc_insts.sh <tag> [knot]      -> gpurun_out/pmc_insts/<tag>.txt
set -u
TAG=$1; FAMILY=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_insts
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SRC="--standin 871316"; [ "$FAMILY" = "knot" ] && SRC="$SRC --knot"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES \
  --output-format csv -d "$OUT/$TAG" -- "$ROOT/ray_tracer_amd/rtx_main" --scene dragon $SRC --width 3840 --height 2160 --mode pruned --repeats 3 --out /tmp/pmc_canvas.ppm > "$OUT/$TAG.log" 2>&1 || { tail -3 "$OUT/$TAG.log"; exit 1; }
python3 - "$OUT/$TAG" > "$OUT/$TAG.txt" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
last = agg[sorted(agg, key=int)[-1]]
for k in sorted(last): print(f"{k:20s} {last[k]:.4g}")
PY
cat "$OUT/$TAG.txt"

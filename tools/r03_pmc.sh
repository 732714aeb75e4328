#!/bin/bash
# Counter passes of the benchmarked frames (profiles/r03_*_pmc.json, profiles/traffic.json): runs on the GPU box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/r03
cp profiles/traffic.json gpurun_out/r03/traffic.json 2>/dev/null || echo '{}' > gpurun_out/r03/traffic.json
one() {  # tag scene tris W H mode family key
  bash tools/pmc_passes.sh $1 $2 $3 $4 $5 $6 $7 > gpurun_out/r03/$1.log 2>&1 || { echo "$1 failed"; tail -3 gpurun_out/r03/$1.log; return 1; }
  python tools/pmc_summary.py gpurun_out/pmc/$1 gpurun_out/r03/$1_pmc.json gpurun_out/r03/traffic.json $8 || return 1
}
one r03_dragon_4k_pruned dragon 871316 3840 2160 pruned blob dragon_3840x2160_pruned || exit 1
one r03_dragon_4k_reference dragon 871316 3840 2160 reference blob dragon_3840x2160_reference || exit 1
one r03_dragon_knot_4k_pruned dragon 871316 3840 2160 pruned knot dragon_knot_3840x2160_pruned || exit 1
one r03_bunny_4k_pruned bunny 144056 3840 2160 pruned blob bunny_3840x2160_pruned || exit 1
one r03_erato_4k_pruned erato 412508 3840 2160 pruned blob erato_3840x2160_pruned || exit 1
one r03_aurelius_4k_pruned aurelius 1704768 3840 2160 pruned blob aurelius_3840x2160_pruned || exit 1
one r03_teapot_4k_pruned teapot 0 3840 2160 pruned blob teapot_3840x2160_pruned || exit 1
echo done

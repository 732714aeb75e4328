#!/bin/bash
# Everything under profiles/r03_*: runs on the GPU box (through gpurun), writes into gpurun_out/r03/.
# usage: tools/r03_profiles.sh [quick]     quick = the default bench line, its kernel trace and the scheduler statistics only
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03
mkdir -p "$OUT"
cd "$ROOT"
B="timeout -k 10 500 python bench.py"
$B > $OUT/bench.json 2> $OUT/bench.err || exit 1
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err) || exit 1
timeout -k 10 200 python tools/sched_stats.py dragon pruned > $OUT/sched_stats.txt 2>&1 || exit 1
timeout -k 10 200 python tools/sched_stats.py dragon reference >> $OUT/sched_stats.txt 2>&1 || exit 1
[ "${1:-}" = "quick" ] && { echo done; exit 0; }
$B --inflight 1 --no-cpu-baseline > $OUT/bench_inflight1.json 2>> $OUT/bench.err || exit 1
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- python3 $ROOT/bench.py --no-cpu-baseline --inflight 1 --no-other-family > $OUT/bench_inflight1_under_rocprof.json 2> $OUT/trace1.err) || exit 1
for s in teapot bunny erato aurelius; do $B --scene $s --no-cpu-baseline --steps 40 > $OUT/bench_$s.json 2>> $OUT/bench.err || exit 1; done
$B --scene aurelius --width 7680 --height 4320 --spp 16 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_aurelius_8k16.json 2>> $OUT/bench.err || exit 1
$B --scene teapot --obj tests/golden/teapot.obj --width 1920 --height 1920 > $OUT/bench_teapot_real_1920.json 2>> $OUT/bench.err || exit 1
timeout -k 10 200 python tools/host_loop_time.py dragon 100 > $OUT/host_loop.txt 2>&1 || exit 1
timeout -k 10 200 python tools/wave_timeline.py dragon pruned > $OUT/wave_timeline.txt 2>&1 || exit 1
echo done

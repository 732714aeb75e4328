#!/bin/bash
# Copies what tools/r03_profiles.sh left under gpurun_out/r03/ into profiles/r03_* (run here, after the gpurun call).
set -eu
cd "$(dirname "$0")/.."
K=$(ls -t gpurun_out/r03/trace/vm/*kernel_trace.csv | head -1); K1=$(ls -t gpurun_out/r03/trace1/vm/*kernel_trace.csv | head -1)
S=$(ls -t gpurun_out/r03/trace/vm/*kernel_stats.csv | head -1); S1=$(ls -t gpurun_out/r03/trace1/vm/*kernel_stats.csv | head -1)
{ sed -n 1,7p profiles/r03_bench_trace_agreement.txt
  python tools/trace_agreement.py "$K1" gpurun_out/r03/bench_inflight1_under_rocprof.json "bench.py --inflight 1 --no-other-family"
  python tools/trace_agreement.py "$K" gpurun_out/r03/bench_under_rocprof.json "bench.py default (four in flight, both families)"; } > /tmp/agree.txt
mv /tmp/agree.txt profiles/r03_bench_trace_agreement.txt
cp "$S" profiles/r03_bench_kernel_stats.csv; cp "$S1" profiles/r03_bench_inflight1_kernel_stats.csv
for f in bench bench_under_rocprof bench_inflight1 bench_inflight1_under_rocprof bench_teapot bench_bunny bench_erato bench_aurelius bench_aurelius_8k16 bench_teapot_real_1920; do cp gpurun_out/r03/$f.json profiles/r03_$f.json; done
cp gpurun_out/r03/sched_stats.txt profiles/r03_sched_stats.txt; cp gpurun_out/r03/host_loop.txt profiles/r03_host_loop.txt; cp gpurun_out/r03/wave_timeline.txt profiles/r03_wave_timeline.txt
python - <<'PY'
import json
d = json.load(open('profiles/r03_bench.json'))
c, r = d['config'], d['roofline']
print('knot', d['value'], 'Mrays/s', d['ms_per_step'], 'ms; alone', c['ms_per_frame_one_at_a_time'], 'batch4', c['ms_per_frame_four_frames_per_launch'],
      'own rays', d['value_in_rays_the_timed_kernel_traced'], 'frac', r['frac'], 'lanes', r['lane_utilisation'], 'valu', r['wave_valu_instructions_per_launch'], r['counter_build'])
o = d['other_family']
print('blob', o['value'], o['ms_per_step'], 'alone', o['frame_ms_latency'], 'batch4', o['ms_per_frame_four_frames_per_launch'], 'frac', o['roofline']['frac'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sample'][-70:])
print('reference walk', d['other_mode'])
for s in ['teapot', 'bunny', 'erato', 'aurelius', 'aurelius_8k16', 'teapot_real_1920']:
    x = json.load(open(f'profiles/r03_bench_{s}.json')); print(s, x['value'], x['ms_per_step'])
PY
tail -4 profiles/r03_host_loop.txt

#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args...]
# kernel-trace + stats pass of a short bench run; prints the per-kernel table.
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o $tag -- python3 bench.py --steps 2 --warmup 1 --no-episode --no-cpu --no-emul "$@" > gpurun_out/prof/${tag}_run.log 2>&1
python3 - <<PY
import csv, json
for r in list(csv.DictReader(open("gpurun_out/prof/${tag}_kernel_stats.csv")))[:7]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.2f} pct {r["Percentage"]}')
for line in open("gpurun_out/prof/${tag}_run.log"):
    if line.startswith('{"metric'):
        d = json.loads(line)
        print('exp/s', round(d['value']), 'ms/ply', round(d['ms_per_step'], 2), 'trunk_ms', round(d['roofline']['avg_launch_ms'], 4), 'frac', round(d['roofline']['frac'], 4))
PY

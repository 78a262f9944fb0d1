#!/bin/bash
# GPU box, end of round 3: everything under profiles/r03_* measured again on the final binary, in two calls
# (usage: tools/measure_r03_final.sh a|b).  Results land under gpurun_out/; copy what is to be judged into profiles/.
export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/pmc
if [ "${1:-a}" = a ]; then
  python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; echo "bench rc=$?"
  tools/measure_r03_evidence.sh benches
  python tools/latency.py gpurun_out/r03_latency.json 2>&1 | grep -v amdgpu.ids
  python tools/upgrades_gain.py > gpurun_out/r03_upgrades.json 2> gpurun_out/upg.err; tail -c 600 gpurun_out/r03_upgrades.json
else
  tools/measure_r03_evidence.sh profiles
  python tools/episode_trace.py 4 > gpurun_out/r03_episode_trace.txt 2>&1; tail -4 gpurun_out/r03_episode_trace.txt
fi

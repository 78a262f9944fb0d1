#!/bin/bash
# GPU box, round 3: the -m gpu suite, the latency path under rocprofv3 (single search, 51-game arena), kernel stats with one
# lane (every kernel alone on the GPU) and the headline bench line.  usage: tools/measure_r03.sh [tag] [suite|nosuite]
export TMPDIR=/tmp
tag=${1:-r03}
mkdir -p gpurun_out/prof
if [ "${2:-suite}" = suite ]; then
  AZ_PARITY_REPORT=$PWD/gpurun_out/${tag}_emulated_trunk_parity.json python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_suite.log 2>&1; rc=$?
  tail -4 gpurun_out/${tag}_suite.log
  [ $rc -ne 0 ] && exit $rc
fi
python3 tools/latency_prof.py search > gpurun_out/${tag}_latency.txt 2>&1
python3 tools/latency_prof.py arena >> gpurun_out/${tag}_latency.txt 2>&1
cat gpurun_out/${tag}_latency.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o ${tag}_lat_search -- python3 tools/latency_prof.py search > gpurun_out/prof/${tag}_lat_search.log 2>&1
python3 tools/kstats.py gpurun_out/prof/${tag}_lat_search_kernel_stats.csv 10
tools/prof.sh ${tag}_f32_engines1 --engines 1
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
print("exp/s", round(d["value"]), "ms/ply", round(d["ms_per_step"], 2), "frac", round(d["roofline"]["frac"], 4), "games/s", d["self_play_games_per_sec"], d["self_play_games_per_sec_steady_state"],
      [(r["kernel"][:6], round(r["avg_launch_ms"] * 1e3, 1)) for r in d["roofline"]["rest"]])
PY

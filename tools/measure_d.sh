#!/bin/bash
# GPU box: the fp32-emulating trunk -- its tests, then bench lines with and without it.
set -o pipefail
python -m pytest tests/test_bf16x3_gpu.py -q -x -s > gpurun_out/r02_bf3_tests.log 2>&1; rc=$?; tail -30 gpurun_out/r02_bf3_tests.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 12 --warmup 3 --no-cpu --trunk bf16x3 > gpurun_out/r02_bench_bf16x3.json 2> gpurun_out/e_b3.log || { tail -20 gpurun_out/e_b3.log; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/r02_bench_bf16x3.json"))
print(round(d["value"]), d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["self_play_games_per_sec"], d["self_play_games_per_sec_steady_state"])
PY

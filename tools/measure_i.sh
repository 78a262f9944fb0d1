#!/bin/bash
# GPU box: the emulated trunk incl. the ResidualBlock variant -- tests, then resnet bench lines with and without it.
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_bf16x3_gpu.py -q -x -s > gpurun_out/r02_bf3_tests.log 2>&1; rc=$?; tail -12 gpurun_out/r02_bf3_tests.log
[ $rc -eq 0 ] || exit $rc
python bench.py --model resnet --sims 800 --steps 4 --no-cpu --steady-games 0 > gpurun_out/r02_bench_resnet.json 2> gpurun_out/e_r1.log || tail -5 gpurun_out/e_r1.log
python bench.py --model resnet --sims 800 --steps 4 --no-cpu --steady-games 0 --trunk bf16x3 > gpurun_out/r02_bench_resnet_bf16x3.json 2> gpurun_out/e_r2.log || tail -5 gpurun_out/e_r2.log
python - <<PY
import json
for f in ("r02_bench_resnet", "r02_bench_resnet_bf16x3"):
    try:
        d = json.load(open("gpurun_out/" + f + ".json"))
        print(f, round(d["value"]), round(d["ms_per_step"], 3), round(1e3 * d["roofline"]["avg_launch_ms"], 1), round(d["roofline"]["frac"], 4), round(d["roofline"]["aggregate"]["frac"], 4), d["self_play_games_per_sec"])
    except Exception as ex:
        print(f, "FAILED", ex)
PY

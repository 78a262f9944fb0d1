#!/bin/bash
# GPU box, start of a session: the whole -m gpu suite, then the headline bench line.  Writes under gpurun_out/.
set -o pipefail
python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r02_tests.log 2>&1; rc=$?; tail -25 gpurun_out/r02_tests.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench.json 2> gpurun_out/e_b.log || { tail -20 gpurun_out/e_b.log; exit 1; }
cat gpurun_out/r02_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['aggregate']['frac'], d['self_play_games_per_sec'], d['self_play_games_per_sec_steady_state'], d['cpu_baseline']['value'])"

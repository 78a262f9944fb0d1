#!/bin/bash
# GPU box: the whole -m gpu suite, __graft_entry__.smoke(), then the headline bench line (dtype f32).  Writes under gpurun_out/.
set -o pipefail
python -m pytest tests -m gpu -q -x --durations=10 > gpurun_out/r02_tests.log 2>&1; rc=$?; tail -16 gpurun_out/r02_tests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench.json 2> gpurun_out/e_b.log || { tail -20 gpurun_out/e_b.log; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/r02_bench.json')); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['aggregate']['frac'], d['self_play_games_per_sec'], d['self_play_games_per_sec_steady_state'], d['cpu_baseline']['value'])"

#!/bin/bash
# GPU box: test suite, 5x5 bench with and without the persistent search kernel, latency table.  Writes under gpurun_out/.
python -m pytest tests -m gpu -q -x > gpurun_out/r02_t5.log 2>&1; tail -8 gpurun_out/r02_t5.log
python bench.py --board 5 --win 4 --sims 100 --steps 8 --warmup 2 --no-cpu > gpurun_out/r02_bench_5x5_persist.json 2> gpurun_out/e1.log
AZ_PERSIST=0 python bench.py --board 5 --win 4 --sims 100 --steps 8 --warmup 2 --no-cpu > gpurun_out/r02_bench_5x5_lockstep.json 2> gpurun_out/e2.log
python tools/latency.py gpurun_out/r02_latency.json > gpurun_out/r02_latency.log 2>&1
python - <<PY
import json
for f in ("r02_bench_5x5_persist", "r02_bench_5x5_lockstep"):
    try:
        d = json.load(open("gpurun_out/" + f + ".json"))
        print(f, round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["aggregate"]["frac"], d["self_play_games_per_sec"],
              d["self_play_games_per_sec_steady_state"], d["config"]["search_kernel"])
    except Exception as ex:
        print(f, "FAILED", ex)
print(open("gpurun_out/r02_latency.log").read())
PY
tail -5 gpurun_out/e1.log gpurun_out/e2.log

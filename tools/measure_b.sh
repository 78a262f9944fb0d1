#!/bin/bash
# GPU box: validation of the newest kernels, 5x5 persistent vs lock-step at three occupancies, search upgrades, headline bench.
set -o pipefail
python -m pytest tests/test_shims_gpu.py tests/test_persistent_gpu.py tests/test_search_upgrades_gpu.py -x -q > gpurun_out/r02_t6.log 2>&1; tail -5 gpurun_out/r02_t6.log
for slots in 1024 512 256; do
  python bench.py --board 5 --win 4 --sims 100 --slots $slots --steps 8 --warmup 2 --no-cpu --steady-games 0 > gpurun_out/r02_b5_persist_$slots.json 2> gpurun_out/e.log
  AZ_PERSIST=0 python bench.py --board 5 --win 4 --sims 100 --slots $slots --steps 8 --warmup 2 --no-cpu --steady-games 0 > gpurun_out/r02_b5_lockstep_$slots.json 2>> gpurun_out/e.log
done
python tools/upgrades_gain.py > gpurun_out/r02_upgrades.json 2> gpurun_out/e_up.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench.json 2> gpurun_out/e_b.log
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r02_b5_*.json")) + ["gpurun_out/r02_bench.json"]:
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 3), round(d["roofline"]["frac"], 4), round(d["roofline"]["aggregate"]["frac"], 4),
              d["self_play_games_per_sec"], d.get("self_play_games_per_sec_steady_state"))
    except Exception as ex:
        print(f, "FAILED", ex)
try:
    u = json.load(open("gpurun_out/r02_upgrades.json"))
    for r in u["eval_cache"]: print(r)
    for r in u["virtual_loss"]: print(r)
except Exception as ex:
    print("upgrades FAILED", ex); print(open("gpurun_out/e_up.log").read()[-2000:])
PY

#!/bin/bash
# GPU box: games/s of the 1024-game 15x15 episode against the active-slot threshold of the low-latency (tile-split) trunk
export TMPDIR=/tmp
for sm in ${@:-32 64 128 256}; do
  AZ_SPLIT_MAX=$sm python3 bench.py --steps 4 --warmup 1 --no-cpu --steady-games 0 > gpurun_out/sweep_$sm.json 2>/dev/null
  python3 - <<PY
import json
d = json.load(open("gpurun_out/sweep_$sm.json"))
print("split_max $sm: exp/s", round(d["value"]), "ms/ply", round(d["ms_per_step"], 2), "episode s", round(d["episode"]["seconds"], 3), "games/s", round(d["self_play_games_per_sec"], 2), "fc us", round(d["roofline"]["rest"][1]["avg_launch_ms"] * 1e3, 1))
PY
done

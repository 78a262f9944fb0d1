#!/bin/bash
# GPU box: the 4-lane pipeline's ply time against the shape of k_fc (1 = one tile per workgroup, 8 = eight in four rounds), interleaved
export TMPDIR=/tmp
for rep in 1 2; do for sh in ${@:-8 1}; do
  AZ_FC_SHAPE=$sh python3 bench.py --steps 10 --warmup 2 --no-cpu --no-episode > gpurun_out/abfc_$sh.json 2>/dev/null
  python3 - <<PY
import json
d = json.load(open("gpurun_out/abfc_$sh.json"))
print("fc shape $sh: exp/s", round(d["value"]), "ms/ply", round(d["ms_per_step"], 2), "trunk us", round(d["roofline"]["avg_launch_ms"] * 1e3, 2), "fc us", round(d["roofline"]["rest"][1]["avg_launch_ms"] * 1e3, 1), "agg", round(d["roofline"]["aggregate"]["frac"], 4))
PY
done; done

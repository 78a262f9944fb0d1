#!/bin/bash
# usage (GPU box): tools/pmc.sh <tag> "<COUNTER ...>"   -- one PMC pass (with --kernel-trace only), per-kernel means printed
export TMPDIR=/tmp
tag=$1; ctrs=$2
mkdir -p gpurun_out/pmc
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc -o $tag -- python3 bench.py --steps 1 --warmup 0 --no-episode --no-cpu --pmc-run --engines 1 > gpurun_out/pmc/${tag}_run.log 2>&1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/pmc/${tag}_counter_collection.csv")))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "k_trunk" in k or "k_fc" in k or "k_step" in k:
        print(k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in d.items()})
PY

#!/bin/bash
# usage (GPU box): tools/pmc.sh <tag> "<COUNTER ...>" [bench args...]   -- one PMC pass (with --kernel-trace only), per-kernel means printed
export TMPDIR=/tmp
tag=$1; ctrs=$2; shift; shift
mkdir -p gpurun_out/pmc
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc -o $tag -- python3 bench.py --steps 1 --warmup 0 --no-episode --no-cpu --pmc-run --engines 1 "$@" > gpurun_out/pmc/${tag}_run.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc/${tag}_counter_collection.csv

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/calib
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/calib -o run -- python3 bench.py --steps 2 --warmup 1 --no-episode --no-cpu > gpurun_out/calib/bench.json 2> gpurun_out/calib/err.log || exit 1
python3 tools/calib_from_trace.py gpurun_out/calib/run_kernel_trace.csv gpurun_out/calib/bench.json gpurun_out/calib/summary.json
rm -f gpurun_out/calib/run_kernel_trace.csv

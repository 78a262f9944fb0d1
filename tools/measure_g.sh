#!/bin/bash
# GPU box: the emulated trunk after a kernel change -- parity tests, bench lines (1 and 4 lanes), phase stamps.
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_bf16x3_gpu.py -q -x -s > gpurun_out/r02_bf3_tests.log 2>&1; rc=$?; tail -6 gpurun_out/r02_bf3_tests.log
[ $rc -eq 0 ] || exit $rc
for eng in 1 4; do
  python bench.py --steps 6 --warmup 2 --no-cpu --no-episode --trunk bf16x3 --engines $eng 2>> gpurun_out/e_g.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('engines=$eng: exp/s', round(d['value']), 'ms/ply', round(d['ms_per_step'],2), 'trunk_us', round(1e3*d['roofline']['avg_launch_ms'],2), 'boards', d['roofline']['boards_per_launch'])"
done
AZ_ENGINE_LIB=$PWD/alphazero-piskvorky_amd/libaz_engine_stamps.so python tools/stamps.py 15 bf16x3 > gpurun_out/r02_stamps_bf3.txt 2>&1; cat gpurun_out/r02_stamps_bf3.txt

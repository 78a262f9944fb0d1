#!/bin/bash
# GPU box: A/B of the fp32-emulating trunk's experiment builds, its phase stamps, rocprofv3 stats and PMC passes.
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_bf16x3_gpu.py -q -x > gpurun_out/r02_bf3_tests.log 2>&1; rc=$?; tail -6 gpurun_out/r02_bf3_tests.log
[ $rc -eq 0 ] || exit $rc
for v in base il1 il3 ntw2; do
  if [ "$v" = base ]; then unset AZ_ENGINE_LIB; else export AZ_ENGINE_LIB=$PWD/alphazero-piskvorky_amd/libaz_engine_$v.so; fi
  python -m pytest tests/test_bf16x3_gpu.py -q -x -k "net_outputs and 15" > gpurun_out/ab_$v.log 2>&1 || { echo "$v: parity FAILED"; tail -5 gpurun_out/ab_$v.log; continue; }
  for eng in 1 4; do
    python bench.py --steps 6 --warmup 2 --no-cpu --no-episode --trunk bf16x3 --engines $eng 2>> gpurun_out/ab_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v engines=$eng: exp/s', round(d['value']), 'ms/ply', round(d['ms_per_step'],2), 'trunk_us', round(1e3*d['roofline']['avg_launch_ms'],2), 'boards', d['roofline']['boards_per_launch'])"
  done
done
export AZ_ENGINE_LIB=$PWD/alphazero-piskvorky_amd/libaz_engine_stamps.so
python tools/stamps.py 15 bf16x3 > gpurun_out/r02_stamps_bf3.txt 2>&1; cat gpurun_out/r02_stamps_bf3.txt
python tools/stamps.py 15 > gpurun_out/r02_stamps_f32.txt 2>&1; cat gpurun_out/r02_stamps_f32.txt
unset AZ_ENGINE_LIB

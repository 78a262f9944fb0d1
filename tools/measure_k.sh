#!/bin/bash
# GPU box: the f32 trunk after a change -- parity on the built sizes, bench lines with 1 and 4 lanes, phase stamps.
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_engine_gpu.py tests/test_resnet_gpu.py tests/test_persistent_gpu.py -q -x -k "not 6-4-60 and not 7-5-40 and not 3-3-30 and not 4-3-1" > gpurun_out/r02_k_tests.log 2>&1; rc=$?; tail -4 gpurun_out/r02_k_tests.log
[ $rc -eq 0 ] || exit $rc
for eng in 1 4; do
  python bench.py --steps 8 --warmup 2 --no-cpu --no-episode --engines $eng 2>> gpurun_out/e_k.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('engines=$eng: exp/s', round(d['value']), 'ms/ply', round(d['ms_per_step'],2), 'trunk_us', round(1e3*d['roofline']['avg_launch_ms'],2), 'frac', round(d['roofline']['frac'],4), 'agg', round(d['roofline']['aggregate']['frac'],4))"
done
AZ_ENGINE_LIB=$PWD/alphazero-piskvorky_amd/libaz_engine_stamps.so python tools/stamps.py 15 > gpurun_out/r02_stamps_f32_new.txt 2>&1; cat gpurun_out/r02_stamps_f32_new.txt

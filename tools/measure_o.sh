#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_emulated_trunk_gpu.py tests/test_leaf_symmetry_gpu.py -q -x > gpurun_out/r02_emul_tests.log 2>&1; rc=$?; tail -4 gpurun_out/r02_emul_tests.log
[ $rc -eq 0 ] || exit $rc
for tr in bf16x3 f16x2; do
  for eng in 1 4; do
    python bench.py --steps 6 --warmup 2 --no-cpu --no-episode --trunk $tr --engines $eng 2>> gpurun_out/e_m.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tr engines=$eng: exp/s', round(d['value']), 'ms/ply', round(d['ms_per_step'],2), 'trunk_us', round(1e3*d['roofline']['avg_launch_ms'],2), 'frac', round(d['roofline']['frac'],3))"
  done
  AZ_ENGINE_LIB=$PWD/alphazero-piskvorky_amd/libaz_engine_stamps.so python tools/stamps.py 15 $tr 2>&1 | tail -8
done
python bench.py --model resnet --sims 800 --steps 4 --no-cpu --no-episode --trunk f16x2 2>> gpurun_out/e_m.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('resnet f16x2: exp/s', round(d['value']), 'ms/ply', round(d['ms_per_step'],2), 'trunk_us', round(1e3*d['roofline']['avg_launch_ms'],2))"

#!/bin/bash
# GPU box: interleaved A/B of experiment builds in the 4-lane pipeline (libaz_engine_<name>.so): tools/ab_pipe.sh name1 name2 ...
export TMPDIR=/tmp
for rep in 1 2; do for v in "$@"; do
  AZ_ENGINE_LIB=$PWD/alphazero-piskvorky_amd/libaz_engine_$v.so python3 bench.py --steps 10 --warmup 2 --no-cpu --no-emul --no-episode 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v: exp/s', round(d['value']), 'ms/ply', round(d['ms_per_step'],2), 'trunk us', round(d['roofline']['avg_launch_ms']*1e3,2), 'step us', round(d['roofline']['rest'][0]['avg_launch_ms']*1e3,1), 'fc us', round(d['roofline']['rest'][1]['avg_launch_ms']*1e3,1), 'agg', round(d['roofline']['aggregate']['frac'],4))"
done; done

#!/bin/bash
# A/B of experiment builds on the small-board configs: tools/ab_small.sh <board> <win> <sims> <slots> name...
n=$1; k=$2; s=$3; b=$4; shift 4
for v in "$@"; do
  if [ "$v" = base ]; then unset AZ_ENGINE_LIB; else export AZ_ENGINE_LIB=$PWD/alphazero-piskvorky_amd/libaz_engine_$v.so; fi
  for e in 1 4; do
    timeout -k 10 100 python bench.py --board $n --win $k --sims $s --slots $b --engines $e --steps 4 --warmup 1 --no-cpu --no-episode | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'engines', d['config']['engines_per_gpu'], 'exp/s', round(d['value']), 'ms/ply', round(d['ms_per_step'], 2), 'frac', round(d['roofline']['frac'], 3), 'trunk_ms', round(d['roofline']['avg_launch_ms'], 4))" || exit 1
  done
done

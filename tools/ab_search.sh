#!/bin/bash
# GPU box: A/B of experiment builds of the persistent search kernel (5x5, 1024 games, 100 simulations, one lane):
# tools/ab_search.sh name...   (libaz_engine_<name>.so; stamps builds print the phase table instead)
export TMPDIR=/tmp
for v in "$@"; do
  export AZ_ENGINE_LIB=$PWD/alphazero-piskvorky_amd/libaz_engine_$v.so
  case $v in
    *s) python tools/stamps_search.py 5 4 100 2>&1 | grep -v amdgpu.ids ;;
    *) for rep in 1 2; do python bench.py --board 5 --win 4 --sims 100 --slots 1024 --engines 1 --steps 6 --warmup 2 --no-cpu --no-episode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'exp/s', round(d['value']), 'ms/ply', round(d['ms_per_step'], 3), 'frac', round(d['roofline']['frac'], 3))"; done ;;
  esac
done

#!/bin/bash
# usage (GPU box): tools/ab.sh name1 name2 ...   -- A/B experiment builds (libaz_engine_<name>.so; "base" = default), engines=1
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset AZ_ENGINE_LIB; else export AZ_ENGINE_LIB=$PWD/alphazero-piskvorky_amd/libaz_engine_$v.so; fi
  echo "== $v"; tools/prof.sh ab_$v --engines 1 2>&1 | grep -E "k_trunk|exp/s"
  python bench.py --steps 3 --warmup 1 --no-cpu --no-episode 2>&1 | grep metric | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   engines=4: exp/s', round(d['value']), 'ms/ply', round(d['ms_per_step'],2))"
done

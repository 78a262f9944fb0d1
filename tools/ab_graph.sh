#!/bin/bash
# A/B: hipGraph ply replay (default) vs kernel-by-kernel launches (AZ_GRAPH=0) on the three board sizes + latency path
set -e
for g in 1 0; do
  echo "== AZ_GRAPH=$g"
  AZ_GRAPH=$g timeout -k 10 120 python bench.py --board 5 --win 4 --sims 100 --steps 40 --warmup 2 --no-cpu --no-episode | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('5x5', round(d['value']), d['ms_per_step'])"
  AZ_GRAPH=$g timeout -k 10 120 python bench.py --board 9 --win 5 --sims 200 --slots 4096 --steps 8 --warmup 1 --no-cpu --no-episode | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('9x9', round(d['value']), d['ms_per_step'])"
  AZ_GRAPH=$g timeout -k 10 120 python bench.py --steps 6 --warmup 1 --no-cpu --no-episode | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('15x15', round(d['value']), d['ms_per_step'])"
  AZ_GRAPH=$g timeout -k 10 120 python tools/latency.py 2>&1 | tail -4
done

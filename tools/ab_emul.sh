export TMPDIR=/tmp
for rep in 1 2; do for v in ${@:-emb emp}; do
  export AZ_ENGINE_LIB=$PWD/alphazero-piskvorky_amd/libaz_engine_$v.so
  for tr in bf16x3 f16x2; do
    python3 bench.py --steps 8 --warmup 2 --no-cpu --no-episode --trunk $tr 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $tr: exp/s', round(d['value']), 'ms/ply', round(d['ms_per_step'],2), 'trunk us', round(d['roofline']['avg_launch_ms']*1e3,2))"
  done
  python3 bench.py --steps 4 --warmup 1 --no-cpu --no-episode --model resnet --sims 800 --trunk f16x2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v resnet f16x2: exp/s', round(d['value']), 'ms/ply', round(d['ms_per_step'],2), 'trunk us', round(d['roofline']['avg_launch_ms']*1e3,2))"
done; done

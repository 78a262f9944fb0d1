#!/bin/bash
# GPU box: the round's profiles -- rocprofv3 kernel stats and PMC passes per trunk mode, phase stamps (needs the stamps
# variant: make -C alphazero-piskvorky_amd/csrc variant NAME=stamps EXTRA=-DAZ_STAMPS ONLY=15), the bench lines of the opt-in
# modes and of the smaller configs.  Everything lands under gpurun_out/; copy what is to be judged into profiles/.
# usage: tools/measure_evidence.sh [profiles|benches]   (two calls: each stays well inside a 20-minute gpurun limit)
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/pmc
part=${1:-profiles}
if [ "$part" = profiles ]; then
for tr in f32 bf16x3 f16x2; do
  tools/pmc.sh r02_${tr}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --trunk $tr || tail -5 gpurun_out/pmc/r02_${tr}_mfma_run.log
  tools/pmc.sh r02_${tr}_fetch "FETCH_SIZE" --trunk $tr
  tools/pmc.sh r02_${tr}_write "WRITE_SIZE" --trunk $tr
  tools/prof.sh r02_${tr}_engines1 --engines 1 --trunk $tr
  tools/prof.sh r02_${tr}_default --trunk $tr
done
if [ -f alphazero-piskvorky_amd/libaz_engine_stamps.so ]; then
  for tr in f32 bf16x3 f16x2; do
    AZ_ENGINE_LIB=$PWD/alphazero-piskvorky_amd/libaz_engine_stamps.so python tools/stamps.py 15 $tr > gpurun_out/r02_stamps_$tr.txt 2>&1
  done
fi
exit 0
fi
for tr in bf16x3 f16x2; do           # the f32 bench lines are the headline (measure_suite.sh) and the plain runs below
  python bench.py --steps 20 --warmup 5 --no-cpu --trunk $tr > gpurun_out/r02_bench_${tr}.json 2>> gpurun_out/e_ev.log
  python bench.py --board 9 --win 5 --sims 200 --slots 4096 --steps 6 --no-cpu --trunk $tr > gpurun_out/r02_bench_9x9_${tr}.json 2>> gpurun_out/e_ev.log
  python bench.py --board 5 --win 4 --sims 100 --steps 4 --no-cpu --trunk $tr > gpurun_out/r02_bench_5x5_${tr}.json 2>> gpurun_out/e_ev.log
  python bench.py --model resnet --sims 800 --steps 4 --no-cpu --steady-games 0 --trunk $tr > gpurun_out/r02_bench_resnet_${tr}.json 2>> gpurun_out/e_ev.log
done
python bench.py --board 9 --win 5 --sims 200 --slots 4096 --steps 6 --no-cpu > gpurun_out/r02_bench_9x9.json 2>> gpurun_out/e_ev.log
python bench.py --board 5 --win 4 --sims 100 --steps 4 --no-cpu > gpurun_out/r02_bench_5x5.json 2>> gpurun_out/e_ev.log
AZ_PERSIST=0 python bench.py --board 5 --win 4 --sims 100 --steps 4 --no-cpu > gpurun_out/r02_bench_5x5_lockstep.json 2>> gpurun_out/e_ev.log
python bench.py --model resnet --sims 800 --steps 4 --no-cpu --steady-games 0 > gpurun_out/r02_bench_resnet.json 2>> gpurun_out/e_ev.log
python tools/latency.py gpurun_out/r02_latency.json > gpurun_out/r02_latency.log 2>&1
python tools/upgrades_gain.py > gpurun_out/r02_upgrades.json 2> gpurun_out/e_up.log
python tools/episode_trace.py 4 > gpurun_out/r02_episode_trace.txt 2>&1
python - <<PY
import glob, json
for f in sorted(glob.glob("gpurun_out/r02_bench_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 3), round(d["roofline"]["frac"], 4), round(d["roofline"]["aggregate"]["frac"], 4),
              d["self_play_games_per_sec"], d["self_play_games_per_sec_steady_state"])
    except Exception as ex:
        print(f, "FAILED", ex)
PY

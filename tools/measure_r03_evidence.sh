#!/bin/bash
# GPU box, round 3: rocprofv3 kernel stats (one lane = every kernel alone; default = the 4-lane pipeline; the latency path),
# three PMC passes for the f32 trunk, and the bench lines of the other configs / opt-in trunks.  Everything lands under
# gpurun_out/; copy what is to be judged into profiles/.   usage: tools/measure_r03_evidence.sh [profiles|benches]
export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/pmc
part=${1:-profiles}
if [ "$part" = profiles ]; then
  tools/pmc.sh r03_f32_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" || tail -5 gpurun_out/pmc/r03_f32_mfma_run.log
  tools/pmc.sh r03_f32_fetch "FETCH_SIZE"
  tools/pmc.sh r03_f32_write "WRITE_SIZE"
  python3 tools/pmc_summary.py gpurun_out/pmc/r03_f32_mfma_counter_collection.csv gpurun_out/pmc/r03_f32_fetch_counter_collection.csv gpurun_out/pmc/r03_f32_write_counter_collection.csv --json gpurun_out/r03_pmc_f32_summary.json > /dev/null
  tools/prof.sh r03_f32_engines1 --engines 1
  tools/prof.sh r03_f32_default
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r03_lat_search -- python3 tools/latency_prof.py search > gpurun_out/prof/r03_lat_search.log 2>&1
  python3 tools/kstats.py gpurun_out/prof/r03_lat_search_kernel_stats.csv 6
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r03_lat_arena -- python3 tools/latency_prof.py arena > gpurun_out/prof/r03_lat_arena.log 2>&1
  python3 tools/kstats.py gpurun_out/prof/r03_lat_arena_kernel_stats.csv 6
  exit 0
fi
for tr in bf16x3 f16x2; do
  python bench.py --steps 20 --warmup 5 --no-cpu --trunk $tr > gpurun_out/r03_bench_${tr}.json 2>> gpurun_out/e_ev.log
done
python bench.py --board 9 --win 5 --sims 200 --slots 4096 --steps 6 --no-cpu > gpurun_out/r03_bench_9x9.json 2>> gpurun_out/e_ev.log
python bench.py --board 5 --win 4 --sims 100 --steps 4 --no-cpu > gpurun_out/r03_bench_5x5.json 2>> gpurun_out/e_ev.log
python bench.py --model resnet --sims 800 --steps 4 --no-cpu --steady-games 0 > gpurun_out/r03_bench_resnet.json 2>> gpurun_out/e_ev.log
python tools/e2e_selfplay.py > gpurun_out/r03_e2e_selfplay.txt 2>&1
python - <<PY
import glob, json
for f in sorted(glob.glob("gpurun_out/r03_bench_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 3), round(d["roofline"]["frac"], 4), round(d["roofline"]["aggregate"]["frac"], 4),
              d["self_play_games_per_sec"], d["self_play_games_per_sec_steady_state"])
    except Exception as ex:
        print(f, "FAILED", ex)
PY
tail -3 gpurun_out/r03_e2e_selfplay.txt

#!/bin/bash
# GPU box: round-2 evidence -- rocprofv3 kernel stats and PMC passes of the f32 and the bf16x3 trunk, stamps, the small-board benches.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/pmc
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA" | head -40 > gpurun_out/pmc/mfma_counters.txt; wc -l gpurun_out/pmc/mfma_counters.txt
tools/prof.sh r02_f32_engines1 --engines 1
tools/prof.sh r02_bf16x3_engines1 --engines 1 --trunk bf16x3
tools/prof.sh r02_bf16x3_default --trunk bf16x3
tools/pmc.sh r02_bf16x3_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --trunk bf16x3 || tail -5 gpurun_out/pmc/r02_bf16x3_mfma_run.log
tools/pmc.sh r02_bf16x3_fetch "FETCH_SIZE" --trunk bf16x3
tools/pmc.sh r02_bf16x3_write "WRITE_SIZE" --trunk bf16x3
export AZ_ENGINE_LIB=$PWD/alphazero-piskvorky_amd/libaz_engine_stamps.so
python tools/stamps.py 15 bf16x3 > gpurun_out/r02_stamps_bf3.txt 2>&1; cat gpurun_out/r02_stamps_bf3.txt
python tools/stamps.py 15 > gpurun_out/r02_stamps_f32.txt 2>&1; cat gpurun_out/r02_stamps_f32.txt
unset AZ_ENGINE_LIB
python bench.py --board 5 --win 4 --sims 100 --steps 8 --warmup 2 --no-cpu > gpurun_out/r02_bench_5x5.json 2> gpurun_out/e1.log
AZ_PERSIST=0 python bench.py --board 5 --win 4 --sims 100 --steps 8 --warmup 2 --no-cpu > gpurun_out/r02_bench_5x5_lockstep.json 2> gpurun_out/e2.log
python bench.py --board 9 --win 5 --sims 200 --slots 4096 --steps 6 --no-cpu > gpurun_out/r02_bench_9x9.json 2> gpurun_out/e3.log
python - <<PY
import json
for f in ("r02_bench_5x5", "r02_bench_5x5_lockstep", "r02_bench_9x9"):
    try:
        d = json.load(open("gpurun_out/" + f + ".json"))
        print(f, round(d["value"]), round(d["ms_per_step"], 3), round(d["roofline"]["frac"], 4), round(d["roofline"]["aggregate"]["frac"], 4), d["self_play_games_per_sec"],
              d["self_play_games_per_sec_steady_state"], d["config"]["search_kernel"])
    except Exception as ex:
        print(f, "FAILED", ex)
PY
python tools/episode_trace.py 4 > gpurun_out/r02_episode_trace.txt 2>&1; cat gpurun_out/r02_episode_trace.txt

#!/bin/bash
# GPU box: PMC passes and the full bench line of the final emulated trunk, search-upgrade gains, 5x5 at the round-1 settings.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/pmc
tools/pmc.sh r02_bf16x3_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --trunk bf16x3 || tail -5 gpurun_out/pmc/r02_bf16x3_mfma_run.log
tools/pmc.sh r02_bf16x3_fetch "FETCH_SIZE" --trunk bf16x3
tools/pmc.sh r02_bf16x3_write "WRITE_SIZE" --trunk bf16x3
tools/prof.sh r02_bf16x3_engines1 --engines 1 --trunk bf16x3
tools/prof.sh r02_bf16x3_default --trunk bf16x3
python bench.py --steps 20 --warmup 5 --trunk bf16x3 > gpurun_out/r02_bench_bf16x3.json 2> gpurun_out/e_b3.log || tail -5 gpurun_out/e_b3.log
python bench.py --board 5 --win 4 --sims 100 --steps 4 --no-cpu > gpurun_out/r02_bench_5x5_s4.json 2> gpurun_out/e1.log
AZ_PERSIST=0 python bench.py --board 5 --win 4 --sims 100 --steps 4 --no-cpu > gpurun_out/r02_bench_5x5_s4_lockstep.json 2> gpurun_out/e2.log
python bench.py --board 5 --win 4 --sims 100 --steps 4 --no-cpu --trunk bf16x3 > gpurun_out/r02_bench_5x5_s4_bf16x3.json 2> gpurun_out/e3.log
python bench.py --board 9 --win 5 --sims 200 --slots 4096 --steps 6 --no-cpu --trunk bf16x3 > gpurun_out/r02_bench_9x9_bf16x3.json 2> gpurun_out/e4.log
python - <<PY
import json
for f in ("r02_bench_bf16x3", "r02_bench_5x5_s4", "r02_bench_5x5_s4_lockstep", "r02_bench_5x5_s4_bf16x3", "r02_bench_9x9_bf16x3"):
    try:
        d = json.load(open("gpurun_out/" + f + ".json"))
        print(f, round(d["value"]), round(d["ms_per_step"], 3), round(d["roofline"]["frac"], 4), round(d["roofline"]["aggregate"]["frac"], 4), d["self_play_games_per_sec"],
              d["self_play_games_per_sec_steady_state"], d["config"]["search_kernel"][:12])
    except Exception as ex:
        print(f, "FAILED", ex)
PY
python tools/upgrades_gain.py > gpurun_out/r02_upgrades.json 2> gpurun_out/e_up.log; tail -c 1500 gpurun_out/r02_upgrades.json

#!/bin/bash
# GPU box: evidence for the float16 scheme -- PMC passes, kernel stats, the full bench lines, the full-size soaks.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/pmc
for tr in bf16x3 f16x2; do
  tools/pmc.sh r02_${tr}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --trunk $tr || tail -5 gpurun_out/pmc/r02_${tr}_mfma_run.log
  tools/pmc.sh r02_${tr}_fetch "FETCH_SIZE" --trunk $tr
  tools/pmc.sh r02_${tr}_write "WRITE_SIZE" --trunk $tr
  tools/prof.sh r02_${tr}_engines1 --engines 1 --trunk $tr
  tools/prof.sh r02_${tr}_default --trunk $tr
done
python bench.py --steps 20 --warmup 5 --trunk f16x2 > gpurun_out/r02_bench_f16x2.json 2> gpurun_out/e_n1.log || tail -5 gpurun_out/e_n1.log
python bench.py --board 9 --win 5 --sims 200 --slots 4096 --steps 6 --no-cpu --trunk f16x2 > gpurun_out/r02_bench_9x9_f16x2.json 2> gpurun_out/e_n2.log
python bench.py --board 5 --win 4 --sims 100 --steps 4 --no-cpu --trunk f16x2 > gpurun_out/r02_bench_5x5_f16x2.json 2> gpurun_out/e_n3.log
python bench.py --model resnet --sims 800 --steps 4 --no-cpu --steady-games 0 --trunk f16x2 > gpurun_out/r02_bench_resnet_f16x2.json 2> gpurun_out/e_n4.log
python - <<PY
import json
for f in ("r02_bench_f16x2", "r02_bench_9x9_f16x2", "r02_bench_5x5_f16x2", "r02_bench_resnet_f16x2"):
    try:
        d = json.load(open("gpurun_out/" + f + ".json"))
        print(f, round(d["value"]), round(d["ms_per_step"], 3), round(d["roofline"]["frac"], 4), round(d["roofline"]["aggregate"]["frac"], 4), d["self_play_games_per_sec"], d["self_play_games_per_sec_steady_state"])
    except Exception as ex:
        print(f, "FAILED", ex)
PY
python tools/soak_parity.py 128 plain 400 gpurun_out/r02_soak_plain_f16x2.json 15 f16x2 2>&1 | tail -2
python tools/soak_parity.py 32 resnet 800 gpurun_out/r02_soak_resnet_f16x2.json 15 f16x2 2>&1 | tail -2
python tools/soak_parity.py 512 plain 200 gpurun_out/r02_soak_9x9_f16x2.json 9 f16x2 2>&1 | tail -2
python tools/soak_parity.py 2048 plain 100 gpurun_out/r02_soak_5x5_ckpt_f16x2.json 5 ckpt f16x2 2>&1 | tail -2

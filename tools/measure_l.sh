#!/bin/bash
# GPU box: full-size parity soaks of the round's kernels (complete games against the CPU oracle on the host threads).
set -o pipefail
python tools/soak_parity.py 128 plain 400 gpurun_out/r02_soak_plain.json 2>&1 | tail -2
python tools/soak_parity.py 32 resnet 800 gpurun_out/r02_soak_resnet.json 2>&1 | tail -2
python tools/soak_parity.py 128 plain 400 gpurun_out/r02_soak_plain_bf16x3.json 15 bf16x3 2>&1 | tail -2
python tools/soak_parity.py 32 resnet 800 gpurun_out/r02_soak_resnet_bf16x3.json 15 bf16x3 2>&1 | tail -2
python tools/soak_parity.py 512 plain 200 gpurun_out/r02_soak_9x9_bf16x3.json 9 bf16x3 2>&1 | tail -2
python tools/soak_parity.py 2048 plain 100 gpurun_out/r02_soak_5x5_ckpt_bf16x3.json 5 ckpt bf16x3 2>&1 | tail -2

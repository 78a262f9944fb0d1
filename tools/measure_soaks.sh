#!/bin/bash
# GPU box: full-size parity soaks (complete games against the CPU oracle on the box's host threads; about 20 minutes in all).
# usage: tools/measure_soaks.sh [f32|bf16x3|f16x2]   (default f32: every game must be bit-exact)
set -o pipefail
tr=${1:-f32}
if [ "$tr" = f32 ]; then opt=""; tag=""; else opt="$tr"; tag="_$tr"; fi
python tools/soak_parity.py 128 plain 400 gpurun_out/r02_soak_parity_plain$tag.json 15 $opt
python tools/soak_parity.py 32 resnet 800 gpurun_out/r02_soak_parity_resnet$tag.json 15 $opt
python tools/soak_parity.py 512 plain 200 gpurun_out/r02_soak_parity_9x9$tag.json 9 $opt
python tools/soak_parity.py 2048 plain 100 gpurun_out/r02_soak_parity_5x5_ckpt$tag.json 5 ckpt $opt

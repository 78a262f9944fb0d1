#!/bin/bash
# usage (GPU box): tools/ab.sh name1 name2 ...   -- A/B experiment builds (libaz_engine_<name>.so; "base" = default)
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset AZ_ENGINE_LIB; else export AZ_ENGINE_LIB=$PWD/alphazero-piskvorky_amd/libaz_engine_$v.so; fi
  echo "== $v"; tools/prof.sh ab_$v 2>&1 | grep -E "k_trunk|k_fc|k_step|exp/s"
done

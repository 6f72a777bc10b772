#!/bin/bash
# On the GPU box: time the full-grid trunk layers (64 x 128x128 crops) with the production kernel and each ablated variant.
cd "$(dirname "$0")"
for a in 0 ${ABL:-1 2 3 4 5}; do
  if [ $a = 0 ]; then unset G6D_LIB_PATH; else export G6D_LIB_PATH=$PWD/../gen6d_amd/csrc/_abl/libgen6d_w$a.so; fi
  echo "== ablate $a"; SIZES=big REPS=10 python wino_split_probe.py 2>&1 | grep -v amdgpu
done

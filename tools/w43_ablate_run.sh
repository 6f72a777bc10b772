#!/bin/bash
# On the GPU box: time representative layers on the production F(4x4,3x3) kernel and on each ablated variant.
cd "$(dirname "$0")"
for a in 0 ${ABL:-1 2 3 4 5 7}; do
  if [ $a = 0 ]; then unset G6D_LIB_PATH; else export G6D_LIB_PATH=$PWD/../gen6d_amd/csrc/_abl/libgen6d_x$a.so; fi
  echo "== ablate $a"; python w43_probe.py 2>&1 | grep -v amdgpu
done

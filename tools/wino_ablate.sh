#!/bin/bash
# Build ablated variants of the Winograd kernel (profiling only; results are numerically wrong by construction) next to the
# library: gen6d_amd/csrc/_abl/libgen6d_wA.so, A = 1..5 (see WINO_ABLATE in wino_conv.hip).  Run tools/wino_ablate_run.sh on the GPU.
set -e
cd "$(dirname "$0")/../gen6d_amd/csrc"
make -s
mkdir -p _abl
OTHERS=$(ls *.o | grep -v wino_conv.o)
for a in ${ABL:-1 2 3 4 5}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize -DWINO_ABLATE=$a -c wino_conv.hip -o _abl/wino_$a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS _abl/wino_$a.o -o _abl/libgen6d_w$a.so
done

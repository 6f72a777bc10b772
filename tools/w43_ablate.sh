#!/bin/bash
# Build ablated variants of the F(4x4,3x3) kernel (profiling only; results are numerically wrong by construction) next to the library:
# gen6d_amd/csrc/_abl/libgen6d_x<A>.so (see W43_ABLATE in wino43_conv.hip).  Run tools/w43_ablate_run.sh on the GPU box.
set -e
cd "$(dirname "$0")/../gen6d_amd/csrc"
make -s
mkdir -p _abl
OTHERS=$(ls *.o | grep -v wino43_conv.o)
for a in ${ABL:-1 2 3 4 5 7}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize -DW43_ABLATE=$a -c wino43_conv.hip -o _abl/wino43_$a.o &
done
wait
for a in ${ABL:-1 2 3 4 5 7}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS _abl/wino43_$a.o -o _abl/libgen6d_x$a.so
done

#!/bin/bash
# Profiling build of the F(4x4,3x3) kernel with shader-clock stamps at the phase boundaries of the chunk loop (-DW43_TIMING, see
# wino43_conv.hip): gen6d_amd/csrc/_abl/libgen6d_t.so.  On the GPU box: G6D_LIB_PATH=.../libgen6d_t.so python tools/w43_timing.py
set -e
cd "$(dirname "$0")/../gen6d_amd/csrc"
make -s
mkdir -p _abl
OTHERS=$(ls *.o | grep -v wino43_conv.o)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize -DW43_TIMING ${EXTRA} -c wino43_conv.hip -o _abl/wino43_t.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS _abl/wino43_t.o -o _abl/libgen6d_t.so

#!/bin/bash
# Build ablated variants of the conv kernel (profiling only; results are numerically wrong by construction) and time
# one shape with each: tools/ablate.sh "vol mean_embed"
set -e
cd "$(dirname "$0")/.."
for a in ${ABL:-1 2 3 4 5}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DG6D_ABLATE=$a -shared gen6d_amd/csrc/*.hip -o /tmp/libg6d_ablate$a.so
done
echo "== full kernel"; ONLY="$1" python tools/conv_bench.py | grep -v total
for a in ${ABL:-1 2 3 4 5}; do echo "== ablate $a"; G6D_LIB_PATH=/tmp/libg6d_ablate$a.so ONLY="$1" python tools/conv_bench.py | grep -v total; done

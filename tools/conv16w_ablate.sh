#!/bin/bash
# Ablated builds of the halo-patch kernel (timing only, WRONG results): C16W_ABLATE bit 0 = no patch requests after the first slice,
# bit 1 = no filter requests, bit 2 = no fragment reads, bit 3 (ABL=8) = no K loop at all (what a block costs outside it).   bash tools/conv16w_ablate.sh build   (here, cross-compiles)
#                                                            bash tools/conv16w_ablate.sh run [batch] [mode] > out.md   (on the GPU box)
cd "$(dirname "$0")/.."
D=gen6d_amd/csrc/_abl
if [ "$1" = build ]; then
  mkdir -p $D
  OTHERS=$(ls gen6d_amd/csrc/*.o | grep -v conv16_direct.o)
  for a in ${ABL:-1 2 3 4 7}; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DC16W_ABLATE=$a -c gen6d_amd/csrc/conv16_direct.hip -o $D/conv16_abl$a.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS $D/conv16_abl$a.o -o $D/libg6d_c16w_abl$a.so && rm $D/conv16_abl$a.o
  done
else
  echo "## full kernel"; python tools/conv16_bench.py ${2:-16} ${3:-fp16} | cut -d'|' -f2,4,6,11
  for a in ${ABL:-1 2 3 4 7}; do echo "## C16W_ABLATE=$a"; G6D_LIB_PATH=$PWD/$D/libg6d_c16w_abl$a.so python tools/conv16_bench.py ${2:-16} ${3:-fp16} 2>/dev/null | cut -d'|' -f2,4,6,11; done
fi

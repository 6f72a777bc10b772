cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  if [ $v = 0 ]; then unset G6D_LIB_PATH; else export G6D_LIB_PATH=$PWD/gen6d_amd/csrc/_abl/libgen6d_m3nt.so; fi
  echo "== raw nt $v"; BATCH=8 python tools/layer_table.py 2>&1 | grep "mul aff stats\|## selector"
done | tee gpurun_out/c32_m3nt.log

cd $GRAFT_REPO_ROOT/tools
for a in 0 r2 r2s s f2 r3 0; do
  if [ $a = 0 ]; then unset G6D_LIB_PATH; else export G6D_LIB_PATH=$PWD/../gen6d_amd/csrc/_abl/libgen6d_v$a.so; fi
  echo "== variant $a"; REPS=4 python w43_probe.py 2>&1 | grep -v amdgpu
done | tee ../gpurun_out/c5_cache_policy.log

cd $GRAFT_REPO_ROOT
timeout 300 python tools/scratch/aten_ops.py 16 2>&1 | grep -v amdgpu | head -60 | tee gpurun_out/c12_aten16.txt

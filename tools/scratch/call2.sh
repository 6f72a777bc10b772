cd $GRAFT_REPO_ROOT
timeout 300 python tools/scratch/rccl_worker.py > gpurun_out/c2_worker.out 2> gpurun_out/c2_worker.err; echo "worker rc $?"; tail -3 gpurun_out/c2_worker.out; grep -v "^frame\|^$" gpurun_out/c2_worker.err | tail -12

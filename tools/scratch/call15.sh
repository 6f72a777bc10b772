cd $GRAFT_REPO_ROOT
STEPS=10 bash tools/knob_bench.sh "" "" "" 2>&1 | tee gpurun_out/c15_knob.log

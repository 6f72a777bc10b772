cd $GRAFT_REPO_ROOT
STEPS=10 bash tools/knob_bench.sh "conv_pm=2" "" "conv_pm=2" 2>&1 | tee gpurun_out/c27_knob.log

cd $GRAFT_REPO_ROOT
timeout 600 python tools/lowp_selector_sensitivity.py fp16 2>&1 | grep -v amdgpu > gpurun_out/c9_sens_fp16.md; cat gpurun_out/c9_sens_fp16.md

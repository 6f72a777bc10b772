cd $GRAFT_REPO_ROOT
timeout 600 python tools/lowp_selector_sensitivity.py fp16 2>&1 | grep -v amdgpu > gpurun_out/c10_sens_fp16.md; tail -12 gpurun_out/c10_sens_fp16.md
timeout 600 python tools/lowp_selector_sensitivity.py bf16 2>&1 | grep -v amdgpu > gpurun_out/c10_sens_bf16.md; cat gpurun_out/c10_sens_bf16.md

cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_wino43_gpu.py tests/test_networks_gpu.py tests/test_parity_timed_gpu.py -m gpu -q -k "corr2d or test_detector or headline or batched" 2>&1 | tail -4
BATCH=8 timeout 300 python tools/layer_table.py 2>&1 | grep -v amdgpu.ids | sed -n '/detector/,/total/p' | cut -c1-150 > gpurun_out/r04_lt_det_corr7.md; grep -n "corr\|whole\|total" gpurun_out/r04_lt_det_corr7.md
bash tools/knob_bench.sh "" ""

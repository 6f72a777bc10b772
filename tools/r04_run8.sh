cd $GRAFT_REPO_ROOT
BATCH=8 timeout 300 python tools/layer_table.py 2>&1 | grep -v amdgpu.ids | sed -n '/refiner step/,$p' | cut -c1-140 > gpurun_out/r04_lt_featnet_f43.md; head -24 gpurun_out/r04_lt_featnet_f43.md
timeout 600 python -m pytest tests/test_parity_timed_gpu.py tests/test_networks_gpu.py -m gpu -q -k "batched or refiner or multi_query" 2>&1 | tail -3

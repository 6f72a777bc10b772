cd $GRAFT_REPO_ROOT
export G6D_PARITY_LOG=$PWD/gpurun_out/parity_call1.jsonl
(timeout 500 python -m pytest tests/test_rccl_world1_gpu.py -x -q -s 2>&1 | tail -15) > gpurun_out/c1_rccl_test.log; tail -4 gpurun_out/c1_rccl_test.log
timeout 400 python bench.py --shard-refs --gpus 1 --steps 6 --warmup 2 > gpurun_out/c1_bench_shard1.json 2> gpurun_out/c1_bench_shard1.err; echo "shard bench rc $?"; tail -c 1500 gpurun_out/c1_bench_shard1.json | head -c 1500; echo
(G6D_TEST_SWITCHES="knob:w43_map=1" timeout 400 python -m pytest tests/test_wino43_gpu.py -x -q 2>&1 | tail -4) > gpurun_out/c1_w43_map1_tests.log; tail -2 gpurun_out/c1_w43_map1_tests.log
(G6D_TEST_SWITCHES="knob:w43_map=2" timeout 400 python -m pytest tests/test_wino43_gpu.py -x -q 2>&1 | tail -4) > gpurun_out/c1_w43_map2_tests.log; tail -2 gpurun_out/c1_w43_map2_tests.log
STEPS=8 bash tools/knob_bench.sh "" "w43_map=1" "w43_map=2" "" "w43_map=1" 2>&1 | tee gpurun_out/c1_knob_map.log

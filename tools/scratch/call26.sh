cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "position_major or conv_igemm" 2>&1 | tail -3) > gpurun_out/c26_t.log; tail -2 gpurun_out/c26_t.log
STEPS=10 bash tools/knob_bench.sh "" "conv_pm=0" "" 2>&1 | tee gpurun_out/c26_knob.log
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cached --no-chained --no-sweep --lowp "" --no-graph > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cached --no-chained --no-sweep --lowp "" --no-graph > /dev/null 2>&1
python $R/tools/rocpd_pmc.py /tmp/pf/pmc_results.db FETCH_SIZE > $R/gpurun_out/pmc_fetch.md; python $R/tools/rocpd_pmc.py /tmp/pw/pmc_results.db WRITE_SIZE > $R/gpurun_out/pmc_write.md
python $R/tools/pmc_conv_traffic.py /tmp/pf/pmc_results.db /tmp/pw/pmc_results.db 2 $R/gpurun_out/pmc_conv_traffic.json | cut -c1-400

cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "linear_gemv" 2>&1 | tail -4) > gpurun_out/c21_t1.log; tail -2 gpurun_out/c21_t1.log
(timeout 600 python -m pytest tests/test_networks_gpu.py tests/test_parity_timed_gpu.py -x -q -k "refiner or batched_graph" 2>&1 | tail -4) > gpurun_out/c21_t2.log; tail -2 gpurun_out/c21_t2.log
for k in "" "gemv_mfma=0"; do KNOBS="$k" timeout 300 python -c "
import sys, runpy
sys.path.insert(0, 'tools'); import toolenv
sys.argv = ['bench.py', '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-cached', '--no-chained', '--no-sweep', '--lowp', '']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('[$k]', round(d['value'],1), d['parity_vs_reference']['max_rel_diff_row'], {k:(round(v['avg_launch_us'],1), round(v['frac_of_8TBps'],3)) for k,v in d['hbm_kernels'].items()})"; done

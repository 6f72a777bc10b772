cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_lowp_gpu.py -x -q -k "wino or winograd" 2>&1 | tail -3) > gpurun_out/c29_t.log; tail -2 gpurun_out/c29_t.log
STEPS=10 bash tools/knob_bench.sh "" "wino_map=0" "" 2>&1 | tee gpurun_out/c29_knob.log
for k in "" "wino_map=0" ""; do KNOBS="$k" timeout 300 python -c "
import sys, runpy
sys.path.insert(0, 'tools'); import toolenv
sys.argv = ['bench.py', '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-cached', '--no-chained', '--no-sweep', '--lowp', 'fp16']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); l=d['lowp']['fp16']; print('[$k] fp32', round(d['value'],1), 'lowp fp16', round(l['value'],1), l['selector_logits']['ok'], l['parity_vs_reference']['ref_idx_equal'])"; done | tee gpurun_out/c29_lowp.log

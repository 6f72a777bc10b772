# bench.py under launch-policy knobs (tools/toolenv.py: KNOBS="name=value,..."): bash tools/knob_bench.sh "w43_split_gain=1.0" "w43_chunk_us=2.2" ...
# (an empty string = product defaults; "knobs;module.ATTR=0" adds Python-level switches, tools/toolenv.py PYSW).  One line per setting.
cd ${GRAFT_REPO_ROOT:-.}
for k in "$@"; do
  KNOBS="${k%%;*}" PYSW="$( [[ "$k" == *";"* ]] && echo "${k#*;}" )" timeout 300 python -c "
import sys, runpy
sys.path.insert(0, 'tools'); import toolenv
sys.argv = ['bench.py', '--steps', '${STEPS:-10}', '--warmup', '3', '--no-cpu-baseline', '--no-cached', '--no-chained', '--no-sweep', '--lowp', '']
runpy.run_path('bench.py', run_name='__main__')" 2> /tmp/knob_err.log | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        pv = r.get('parity_vs_reference') or {}
        print('knobs [$k]: %.1f images/s, %.2f ms/step, single %.2f ms, wino %.2f ms/step (frac %.3f), conv %.2f ms/step, parity %.1e argmax %s' % (r['value'], r['ms_per_step'], r.get('single_query_ms') or 0, r['roofline']['ms_per_step'], r['roofline']['frac'], r['roofline_conv']['ms_per_step'], pv.get('max_rel_diff_row', -1), pv.get('ref_idx_equal')))
" || tail -3 /tmp/knob_err.log
done

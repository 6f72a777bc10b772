cd $GRAFT_REPO_ROOT
r() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --lowp "" $LANES > gpurun_out/r29_$name.json 2>/dev/null; python -c "
import json
d = json.loads(open('gpurun_out/r29_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'], 2), d['parity_vs_reference']['ref_idx_equal'])"; }
LANES="--lanes 3" r d3a A=1
LANES="--lanes 4" r q8l4a GPU_MAX_HW_QUEUES=8
LANES="--lanes 3" r d3b A=1
LANES="--lanes 4" r q8l4b GPU_MAX_HW_QUEUES=8
LANES="--lanes 3" r q8l3 GPU_MAX_HW_QUEUES=8
LANES="--lanes 5" r q8l5 GPU_MAX_HW_QUEUES=8

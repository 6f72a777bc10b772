cd $GRAFT_REPO_ROOT
KNOBS="conv_pm=0" BATCH=8 python tools/layer_table.py 2>&1 | grep -A45 "## selector" | grep "4x4x\|8x8x" > gpurun_out/c7_pm0.txt
BATCH=8 python tools/layer_table.py 2>&1 | grep -A45 "## selector" | grep "4x4x\|8x8x" > gpurun_out/c7_pm1.txt
paste -d'\n' gpurun_out/c7_pm0.txt gpurun_out/c7_pm1.txt

cd $GRAFT_REPO_ROOT/tools
G6D_WINO_DEBUG=1 REPS=1 python wino_split_probe.py 2>&1 >/dev/null | sort -u > ../gpurun_out/r16_splits.txt
python wino_split_probe.py > ../gpurun_out/r16_default.txt 2>&1
G6D_WINO_SPLIT_MAX=1 python wino_split_probe.py > ../gpurun_out/r16_nosplit.txt 2>&1
G6D_WINO_SPLIT_GAIN=1.0 python wino_split_probe.py > ../gpurun_out/r16_gain1.txt 2>&1
paste ../gpurun_out/r16_default.txt ../gpurun_out/r16_nosplit.txt ../gpurun_out/r16_gain1.txt | grep -v amdgpu

cd $GRAFT_REPO_ROOT
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pk2 -o bench -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --lowp "" --serial > $R/gpurun_out/r21_serial.log 2>&1; python $R/tools/rocpd_stats.py /tmp/pk2/bench_results.db $R/gpurun_out/r21_serial_stats.md 8 | head -3
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pk -o bench -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --lowp "" > $R/gpurun_out/r21_final.log 2>&1; python $R/tools/rocpd_stats.py /tmp/pk/bench_results.db $R/gpurun_out/r21_final_stats.md 8 | head -3
cd $R
python tools/layer_table.py > gpurun_out/r21_layer_table.md 2>&1

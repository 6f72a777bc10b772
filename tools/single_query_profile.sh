# Kernel trace of ONE query at a time (batch 1, one lane, serial eager): launches per query and where a single query's 8.7 ms go
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk1 -o bench -- python $R/bench.py --batch 1 --lanes 1 --steps 4 --warmup 2 --no-cpu-baseline --no-cached --no-chained --no-sweep --lowp "" --serial > $R/gpurun_out/prof_b1.log 2>&1
python $R/tools/rocpd_stats.py /tmp/pk1/bench_results.db $R/gpurun_out/prof_b1_serial_stats.md 4 | tail -3

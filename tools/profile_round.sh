# Round profile: run from the repo root on the GPU box (gpurun).  Raw outputs go to gpurun_out/, tools/assemble_profiles.py rNN turns
# them into the committed summaries under profiles/.  PMC passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md).
R=$PWD; cd /tmp; export TMPDIR=/tmp
B="--no-cpu-baseline --lowp ''"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pk -o bench -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --lowp "" > $R/gpurun_out/prof_final.log 2>&1; python $R/tools/rocpd_stats.py /tmp/pk/bench_results.db $R/gpurun_out/prof_final_stats.md 8 | head -3
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pk2 -o bench -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --lowp "" --serial > $R/gpurun_out/prof_serial.log 2>&1; python $R/tools/rocpd_stats.py /tmp/pk2/bench_results.db $R/gpurun_out/prof_serial_stats.md 8 | head -3
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pmc -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --lowp "" --no-graph > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pmc -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --lowp "" --no-graph > /dev/null 2>&1
python $R/tools/rocpd_pmc.py /tmp/pf/pmc_results.db FETCH_SIZE > $R/gpurun_out/pmc_fetch.md; python $R/tools/rocpd_pmc.py /tmp/pw/pmc_results.db WRITE_SIZE > $R/gpurun_out/pmc_write.md
python $R/tools/pmc_conv_traffic.py /tmp/pf/pmc_results.db /tmp/pw/pmc_results.db 3 $R/gpurun_out/pmc_conv_traffic.json | cut -c1-300
timeout 200 rocprofv3 --kernel-trace --pmc MfmaUtil -d /tmp/pu -o pmc -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --lowp "" --no-graph --serial > /dev/null 2>&1
python $R/tools/rocpd_pmc.py /tmp/pu/pmc_results.db MfmaUtil > $R/gpurun_out/pmc_mfmautil.md; head -8 $R/gpurun_out/pmc_mfmautil.md
cd $R
python tools/layer_table.py > gpurun_out/layer_table.md 2>&1
REPS=10 python tools/conv_bench.py > gpurun_out/convbench_direct.log 2>&1
WINO=1 REPS=10 python tools/conv_bench.py > gpurun_out/convbench_wino.log 2>&1
REPS=10 python tools/trunk_bench.py > gpurun_out/trunk_bench.md 2>&1

# Round profile: run from the repo root on the GPU box (gpurun).  Raw outputs go to gpurun_out/, tools/assemble_profiles.py rNN turns
# them into the committed summaries under profiles/.  PMC passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md).
# Round 4: F(4x4,3x3) kernels in the detector / refiner.  Round 3: bench.py's default is one hipGraph = one batch of 8 queries that share every launch, 2 batches in flight.
R=$PWD; cd /tmp; export TMPDIR=/tmp
B="--no-cpu-baseline --no-cached --lowp ''"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk -o bench -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-cached --no-chained --no-sweep --lowp "" > $R/gpurun_out/prof_final.log 2>&1; python $R/tools/rocpd_stats.py /tmp/pk/bench_results.db $R/gpurun_out/prof_final_stats.md 4 | head -3
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk2 -o bench -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-cached --no-chained --no-sweep --lowp "" --serial > $R/gpurun_out/prof_serial.log 2>&1; python $R/tools/rocpd_stats.py /tmp/pk2/bench_results.db $R/gpurun_out/prof_serial_stats.md 4 | head -3
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cached --no-chained --no-sweep --lowp "" --no-graph > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cached --no-chained --no-sweep --lowp "" --no-graph > /dev/null 2>&1
python $R/tools/rocpd_pmc.py /tmp/pf/pmc_results.db FETCH_SIZE > $R/gpurun_out/pmc_fetch.md; python $R/tools/rocpd_pmc.py /tmp/pw/pmc_results.db WRITE_SIZE > $R/gpurun_out/pmc_write.md
python $R/tools/pmc_conv_traffic.py /tmp/pf/pmc_results.db /tmp/pw/pmc_results.db 2 $R/gpurun_out/pmc_conv_traffic.json | cut -c1-300
timeout 300 rocprofv3 --kernel-trace --pmc MfmaUtil -d /tmp/pu -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cached --no-chained --no-sweep --lowp "" --no-graph --serial > /dev/null 2>&1
python $R/tools/rocpd_pmc.py /tmp/pu/pmc_results.db MfmaUtil > $R/gpurun_out/pmc_mfmautil.md; head -12 $R/gpurun_out/pmc_mfmautil.md
cd $R
BATCH=8 python tools/layer_table.py > gpurun_out/layer_table_b8.md 2>&1
BATCH=1 python tools/layer_table.py > gpurun_out/layer_table_b1.md 2>&1
BATCH=16 LOWP=fp16 python tools/layer_table.py > gpurun_out/layer_table_fp16.md 2>&1
BATCH=16 python tools/layer_table.py > gpurun_out/layer_table_b16.md 2>&1
# round 6: the direct 16-bit convolution family per trunk layer, the sustained MFMA rate, and the LDS-side counters of the three generations
# of the reduced-precision trunk kernel (16-bit Winograd on fp32 activations, per-tap direct kernel, halo-patch direct kernel)
python tools/conv16_bench.py 16 fp16 > gpurun_out/conv16_bench.md 2>/dev/null
tools/ubench/mfma16_peak.bin > gpurun_out/mfma16_peak.md 2>&1
python tools/ubench/conv16w_scaling.py > gpurun_out/conv16w_scaling.md 2>/dev/null
cd /tmp
for c in SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT MfmaUtil; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pl_$c -o pmc -- python $R/tools/conv16_bench.py 16 fp16 > /dev/null 2>&1
done
( echo "# LDS-side counters of the reduced-precision trunk kernels (rocprofv3 --kernel-trace --pmc <one counter per pass> -- python tools/conv16_bench.py 16 fp16;"
  echo "# every kernel runs the same 14 trunk layers 6 times: per-dispatch means are comparable between kernels)"
  for c in SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT MfmaUtil; do echo; echo "## $c"; python $R/tools/rocpd_pmc.py /tmp/pl_$c/pmc_results.db $c | grep "^| kernel\|^|---\|conv16\|wino16\|wino43"; done ) > $R/gpurun_out/pmc_conv16_lds.md
cd $R

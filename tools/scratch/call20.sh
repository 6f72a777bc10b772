cd $GRAFT_REPO_ROOT
export G6D_PARITY_LOG=$PWD/gpurun_out/parity_r05.jsonl; rm -f $G6D_PARITY_LOG
(timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | tail -40) > gpurun_out/c20_tests.log; tail -5 gpurun_out/c20_tests.log

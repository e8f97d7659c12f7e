#!/bin/bash
# round 4, ninth GPU session: prepared W_hh fragments for both cluster recurrences in training; matrix-core BPTT A/B again
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4i
mkdir -p $O
cd $R
SIR_BPTT=5 timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_robustness_gpu.py -x -q -m gpu > $O/tests_bq.log 2>&1 || { tail -40 $O/tests_bq.log; exit 1; }
tail -2 $O/tests_bq.log
for m in 4 5 4 5; do
  SIR_BPTT=$m timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag bptt$m --kernels gru_l > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_bptt_quad.jsonl
done
for k in 6 14 22 30; do
  SIR_BPTT=5 SIR_BQ_DBG=$k timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag bq_dbg$k --kernels bwd_gru_l > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_bptt_quad.jsonl
done

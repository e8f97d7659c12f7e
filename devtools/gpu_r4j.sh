#!/bin/bash
# round 4, tenth GPU session: GRU weight-gradient GEMMs on a side stream beside the half-chip BPTT (SIR_BWD_STREAMS=2)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4j
mkdir -p $O
cd $R
SIR_BWD_STREAMS=3 timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_nccl_gpu.py -x -q -m gpu > $O/tests_s2.log 2>&1 || { tail -40 $O/tests_s2.log; exit 1; }
tail -2 $O/tests_s2.log
for m in 0 1 3 2 0 1 3; do
  SIR_BWD_STREAMS=$m timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag streams$m --kernels none > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_streams2.jsonl
done

#!/bin/bash
# which stage makes the conv1 / conv2 weight-gradient errors of the B = 256 test grow?  (SIR_F16 bit per stage, SIR_BPTT)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/bisect
mkdir -p $O
cd $R
for cfg in "63 4" "63 0" "47 4" "31 4" "59 4" "55 4" "7 4" "0 4"; do
  set -- $cfg
  SIR_F16=$1 SIR_BPTT=$2 timeout -k 10 300 python -m pytest tests/test_train_gpu.py -q -m gpu -s -k "test_training_step_at_bench_batch_256_vs_oracle" > $O/t_$1_$2.log 2>&1
  echo "SIR_F16=$1 SIR_BPTT=$2: $(grep -o "B=256 grad errors: {'conv1.weight': '[^']*', 'bn1.weight': '[^']*', 'bn1.bias': '[^']*', 'conv2.weight': '[^']*'" $O/t_$1_$2.log | head -1) $(tail -1 $O/t_$1_$2.log)"
done

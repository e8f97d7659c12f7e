#!/bin/bash
# round 4: conv1 moment kernel workgroup totals (SIR_C1M; 2048 = the product's)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4v
mkdir -p $O
cd $R
for v in "SIR_C1M=2048" "SIR_C1M=4096" "SIR_C1M=8192" "SIR_C1M=2048" "SIR_C1M=4096" "SIR_C1M=8192"; do
  env $v timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag "$v" --kernels train_conv1 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab2.jsonl
done

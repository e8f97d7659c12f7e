#!/bin/bash
# round 4: does pulling layer 0's saved gates / outputs into the last-level cache beside layer 1's dX shorten layer 0's BPTT?  (SIR_BPTT_TOUCH)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4t
mkdir -p $O
cd $R
for v in 0 1 3 7 5 0 1 3 7; do
  SIR_BPTT_TOUCH=$v timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag touch$v --kernels bwd_gru_l > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_touch.jsonl
done

#!/bin/bash
# round 4: priority of the backward's side stream (SIR_SIDE_PRIO: 0 default, 1 lowest, -1 highest)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4w
mkdir -p $O
cd $R
for v in 0 1 -1 0 1 -1; do
  SIR_SIDE_PRIO=$v timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag "prio$v" --kernels none > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab.jsonl
done

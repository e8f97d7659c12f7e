#!/bin/bash
# A/B of one environment switch on the same box: VAR=name VALUES="0 1" bash devtools/gpu_ab_env.sh  (3 alternating rounds)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for round in 1 2 3; do
for v in $VALUES; do
  env $VAR=$v timeout -k 10 300 python bench.py --steps ${STEPS:-60} --warmup 10 --train-steps ${TRAIN_STEPS:-40} --no-cpu-baseline > gpurun_out/abenv_$v.json 2> gpurun_out/abenv_$v.err || { tail -5 gpurun_out/abenv_$v.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/abenv_$v.json").read().strip().splitlines()[-1])
print("$VAR=$v", "infer", d["value"], "train", d["train"]["value"], d["train"]["ms_per_step"])
PY
done
done

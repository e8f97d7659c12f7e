#!/bin/bash
# A/B of one env switch inside one GPU-box session: ./gpu_ab.sh VAR val_a val_b
# (model/train parity tests under val_b first, then alternating inference-only bench runs)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
VAR=$1; A=$2; B=$3
env $VAR=$B timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py -q -x > gpurun_out/ab_tests.log 2>&1; rc=$?
tail -3 gpurun_out/ab_tests.log
if [ $rc -ne 0 ]; then grep -E "^E|Error" gpurun_out/ab_tests.log | head -20; exit $rc; fi
for v in $A $B $A $B; do
  env $VAR=$v timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --train-steps 10 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -5 gpurun_out/ab_$v.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
print("$VAR=$v", d["value"], d["ms_per_step"], {k: round(x*1e3,1) for k,x in d["kernels_avg_ms"].items() if x}, "train", d.get("train",{}).get("value"))
PY
done

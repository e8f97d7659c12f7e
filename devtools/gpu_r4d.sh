#!/bin/bash
# round 4, fourth GPU session: two-plane f16 weights / B operands, four-k BPTT layout (A/B + knock-outs + parity), HostStager, bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4d
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?
tail -3 $O/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert " $O/gpu_tests.log | head -40; exit 1; fi
SIR_BPTT=4 timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_robustness_gpu.py -x -q -m gpu > $O/tests_k4.log 2>&1 || { tail -30 $O/tests_k4.log; exit 1; }
tail -2 $O/tests_k4.log
for m in 0 4 0 4 37 38 40; do
  SIR_BPTT=$m timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag bptt$m --kernels bwd_gru_l > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_bptt_k4.jsonl
done
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench: infer", d["value"], "ms", d["ms_per_step"], "train", d["train"]["value"], d["train"]["ms_per_step"], "aug", d["train_aug"]["value"])
de=d["train"].get("dropin_epoch", {})
for k in ("dataloader","dataloader_staged","hbm_feature_store","waveform_store"):
    print(k, json.dumps(de.get(k)))
print(json.dumps(d["kernels_avg_ms"]))
print(json.dumps(d["train"]["kernels_avg_ms"]))
PY

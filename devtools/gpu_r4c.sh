#!/bin/bash
# round 4, third GPU session: the backward on f16x3 under the loss scale (GPU suite), per-kernel A/B of SIR_F16, DataLoader probe, bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu -s > $O/gpu_tests.log 2>&1; rc=$?
tail -4 $O/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert " $O/gpu_tests.log | head -40; exit 1; fi
grep -E "train stage errors|grad errors|trajectory" $O/gpu_tests.log | head
for m in 63 7 63 7 15 31 47; do
  SIR_F16=$m timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag f16_$m --kernels bwd_ > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_f16_bwd.jsonl
done
timeout -k 10 400 python devtools/dataloader_probe.py 4096 > $O/dataloader_probe.jsonl 2> $O/probe.err || { tail -20 $O/probe.err; exit 1; }
cat $O/dataloader_probe.jsonl
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench: infer", d["value"], "ms", d["ms_per_step"], "train", d["train"]["value"], d["train"]["ms_per_step"], "aug", d["train_aug"]["value"])
print(json.dumps(d["train"].get("dropin_epoch", {}).get("dataloader")))
print(json.dumps(d["train"]["kernels_avg_ms"]))
PY

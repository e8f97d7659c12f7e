#!/bin/bash
# round 4, fifth GPU session: wino2 with the dedicated exchange area (parity + bench_conv), staged hand-over probe, short bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?
tail -3 $O/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert " $O/gpu_tests.log | head -40; exit 1; fi
timeout -k 10 400 speech-intent-recognizer_amd/lib/bench_conv wino2 > $O/bench_conv_wino2.txt 2>&1 || { tail -20 $O/bench_conv_wino2.txt; exit 1; }
grep -E "^conv|f16x3" $O/bench_conv_wino2.txt | head -40
timeout -k 10 400 python devtools/dataloader_probe.py 2048 staged_only,staged_train,staged_train_side_stream,none > $O/probe2.jsonl 2> $O/probe2.err || { tail -20 $O/probe2.err; exit 1; }
cat $O/probe2.jsonl
timeout -k 10 600 python bench.py --no-dropin > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench: infer", d["value"], "ms", d["ms_per_step"], "serial", d["single_stream"]["ms_per_step"], "train", d["train"]["value"], d["train"]["ms_per_step"], "aug", d["train_aug"]["value"])
print(json.dumps(d["kernels_avg_ms"]))
print(json.dumps(d["train"]["kernels_avg_ms"]))
PY

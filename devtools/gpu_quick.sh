#!/bin/bash
# quick GPU check: model/train/pipeline parity tests, then one bench run (prints the kernel table)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_features_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py tests/test_pipeline_gpu.py -q -x > gpurun_out/quick_tests.log 2>&1; rc=$?
tail -3 gpurun_out/quick_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" gpurun_out/quick_tests.log | head -20; exit $rc; fi
timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline $BENCH_ARGS > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err || { tail -5 gpurun_out/quick_bench.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/quick_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: round(x*1e3,1) for k,x in d["kernels_avg_ms"].items()}, "train", d.get("train",{}).get("value"), d.get("train",{}).get("ms_per_step"))
PY
